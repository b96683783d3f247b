"""The pure-C++ client of the C ABI (examples/rtr_app.cpp over csrc/host/application.hpp): the headless counterpart
of the reference's main() + Application::run()."""
import os
import subprocess

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "realtimeraytracer_amd", "rtr_app")


def test_app_reports_errors_like_reference_main(tmp_path):
    # reference src/main.cpp:12-15: exception -> message on stderr, EXIT_FAILURE
    r = subprocess.run([APP, "/nonexistent/scene.obj", "", str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: " in r.stderr and "Cannot open file" in r.stderr
    r = subprocess.run([APP], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr


def test_app_without_gpu_fails_loudly(tmp_path, scene_cache):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    obj, d = scenes.write_cornell(scene_cache)
    r = subprocess.run([APP, obj, d, str(tmp_path / "o.ppm"), "32", "32"], capture_output=True, text=True)
    assert r.returncode == 1 and "RTR_ERR_NO_DEVICE" in r.stderr and not os.path.exists(tmp_path / "o.ppm")


@pytest.mark.gpu
def test_app_output_equals_harness_render(tmp_path, scene_cache, gpu_ctx):
    obj, d = scenes.write_cornell(scene_cache)
    out = str(tmp_path / "cornell.ppm")
    r = subprocess.run([APP, obj, d, out, "200", "120", "2", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline().split() == [b"P6", b"200", b"120", b"255"]
        rgb = np.frombuffer(f.read(), np.uint8).reshape(120, 200, 3)
    s = scenes.cornell_box(200, 120)
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, 200, 120)
    api.render(scene, s.camera, s.scene_info(0), api.make_params(200, 120, spp=2), frame)
    bgra = frame.download().view(np.uint8).reshape(120, 200, 4)
    assert np.array_equal(rgb, bgra[:, :, [2, 1, 0]])
