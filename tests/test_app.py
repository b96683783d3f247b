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


@pytest.mark.gpu
def test_app_presents_the_reference_frame_under_scripted_input(tmp_path, scene_cache, gpu_ctx, oracle):
    """The frame the reference presents (application.cppm:391-457: five ray-gen images at 4 spp, four a-trous rounds, combine) from the
    C++ application with the SHIPPED LTC tables and three frames of scripted input (W, D, then T), against the same frame made
    through the Python harness — and that one against the oracle, all eight images."""
    W, H = 200, 120
    obj, d = scenes.write_cornell(scene_cache)
    out = str(tmp_path / "present.ppm")
    r = subprocess.run([APP, obj, d, out, str(W), str(H), "4", "3", "present", "keys=W,D,T"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline().split() == [b"P6", str(W).encode(), str(H).encode(), b"255"]
        rgb = np.frombuffer(f.read(), np.uint8).reshape(H, W, 3)
    from realtimeraytracer_amd import host
    s = scenes.cornell_box(W, H, ltc=scenes.shipped_ltc())
    cam = host.Camera(40.0, (278, 273, -800), (278, 273, 0), (0, 1, 0), W, H)
    for keys in ("W", "D", "T"):
        cam.applyInput(keys)
    camera = cam.getGPUData()
    assert tuple(camera.position[:]) != (278.0, 273.0, -800.0)
    info = host.scene_info(2, s.num_lights, tuple(camera.position[:]))
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, W, H, 0xff)
    p = api.make_params(W, H, spp=4, images=A.IMAGES_RAYGEN5)
    api.render(scene, camera, info, p, frame)
    ref = oracle.render(s.desc, camera, info, p, bvh=scene.export_bvh(), images=A.IMAGES_RAYGEN5, threads=16)
    for which in (0, 1, 2, 6, 7):
        assert np.array_equal(frame.download(which), ref.images[which]), f"ray-gen image {which} with the shipped LTC tables"
    frame.denoise_combine(4)
    refd = oracle.denoise_combine(*(ref.images[k] for k in (0, 1, 2, 6, 7)), iterations=4)
    fin = frame.download(A.IMAGE_FINAL)
    assert np.array_equal(fin, refd[A.IMAGE_FINAL])
    assert np.array_equal(rgb, fin.view(np.uint8).reshape(H, W, 4)[:, :, [2, 1, 0]])
    assert rgb.std() > 10
