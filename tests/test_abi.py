"""C-ABI surface: the shared library loads, exports every symbol include/rtr.h declares, the POD layouts
match the reference's, and nothing in the product routes through the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

from realtimeraytracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    lib = A.hip_lib()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"librtr_hip.so does not export {n}"
        assert n in A.RTR_SYMBOLS, f"{n} declared in include/rtr.h but not bound in _abi.RTR_SYMBOLS"
    for n in A.RTR_SYMBOLS:
        assert n in names, f"{n} bound but not declared in include/rtr.h"


def test_mgpu_header_symbols_are_exported_and_bound():
    """include/rtr_mgpu.h <-> librtr_mgpu.so <-> _abi.MGPU_SYMBOLS (load + symbols only: no compute without a GPU)."""
    text = open(os.path.join(ROOT, "include", "rtr_mgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rtr_mgpu_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 12
    lib = A.mgpu_lib()
    for n in names:
        assert hasattr(lib, n), f"librtr_mgpu.so does not export {n}"
        assert n in A.MGPU_SYMBOLS, f"{n} declared in include/rtr_mgpu.h but not bound in _abi.MGPU_SYMBOLS"
    for n in A.MGPU_SYMBOLS:
        assert n in names, f"{n} bound but not declared in include/rtr_mgpu.h"
    out = subprocess.check_output(["nm", "-D", "--defined-only", A.LIB_MGPU_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported
    ldd = subprocess.check_output(["ldd", A.LIB_MGPU_PATH]).decode()
    assert "librccl" in ldd and "librtr_hip" in ldd and "oracle" not in ldd
    assert C.sizeof(A.rtr_mgpu_info) == 56


def test_mgpu_rejects_bad_arguments_and_missing_device():
    import torch
    lib = A.mgpu_lib()
    h = A.VP()
    assert lib.rtr_mgpu_create(None, 0, 1, C.byref(h)) == -1
    assert lib.rtr_mgpu_render(None, None, None, None) == -1
    assert lib.rtr_mgpu_wait(None, 0) == -1
    if not torch.cuda.is_available():
        devs = (C.c_int * 1)(0)
        assert lib.rtr_mgpu_create(devs, 1, 1, C.byref(h)) == -3 and not h.value     # RTR_ERR_NO_DEVICE, never a CPU path
        assert b"no CPU fallback" in lib.rtr_mgpu_last_error()


def test_exports_are_c_linkage():
    out = subprocess.check_output(["nm", "-D", "--defined-only", A.LIB_HIP_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in _declared_functions():
        assert n in exported


def test_pod_layouts_match_reference():
    # Vertex 48, GPUCameraData 64, SceneInfo 32, GPUObjectInfo 80, GPUAreaLightInfo 96, DenoisingInfo 24
    assert C.sizeof(A.RtrVertex) == 48 and A.RtrVertex.normal.offset == 16 and A.RtrVertex.uv.offset == 32
    assert C.sizeof(A.RtrCameraData) == 64 and A.RtrCameraData.topLeftViewportCorner.offset == 16
    assert A.RtrCameraData.horizontalViewportDelta.offset == 32 and A.RtrCameraData.verticalViewportDelta.offset == 48
    assert C.sizeof(A.RtrSceneInfo) == 32 and A.RtrSceneInfo.camPosition.offset == 16
    assert C.sizeof(A.RtrObjectInfo) == 80
    for f, off in (("usesColorMap", 16), ("usesOpacityMap", 28), ("colorIndex", 32), ("opacityIndex", 44), ("color", 48), ("specular", 64), ("metallic", 68)):
        assert getattr(A.RtrObjectInfo, f).offset == off
    assert C.sizeof(A.RtrAreaLightInfo) == 96
    for f, off in (("intensity", 12), ("vertexOffset", 16), ("indexOffset", 20), ("numTriangles", 24), ("isTwoSided", 28), ("transform", 32)):
        assert getattr(A.RtrAreaLightInfo, f).offset == off
    assert C.sizeof(A.RtrBvhNode) == 32 and C.sizeof(A.RtrBvhGrid) == 32 and C.sizeof(A.RtrBvhTri) == 48


def test_abi_version_and_status_strings():
    lib = A.hip_lib()
    assert lib.rtr_abi_version() == 3
    assert lib.rtr_status_string(0) == b"RTR_OK"
    assert lib.rtr_status_string(-3) == b"RTR_ERR_NO_DEVICE"
    assert lib.rtr_status_string(-6) == b"RTR_ERR_BVH_TOO_DEEP"


def test_shard_rows():
    lib = A.hip_lib()
    assert lib.rtr_shard_rows(1080, 8, 1) == 1080
    assert lib.rtr_shard_rows(1080, 8, 8) == 136      # 17 bands -> 136 rows
    assert lib.rtr_shard_rows(2160, 8, 8) == 272
    assert lib.rtr_shard_rows(1080, 0, 0) == 1080
    assert lib.rtr_shard_rows(100, 8, 3) == 40


def test_no_device_fails_loudly():
    """On a box without a GPU the product must refuse, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = A.hip_lib()
    h = A.VP()
    rc = lib.rtr_ctx_create(0, C.byref(h))
    assert rc == -3 and not h.value
    assert b"no CPU fallback" in lib.rtr_last_error()


def test_null_arguments_are_rejected():
    lib = A.hip_lib()
    assert lib.rtr_scene_get_stats(None, None) == -1
    assert lib.rtr_frame_download(None, 1, None, 0) == -1
    assert lib.rtr_render(None, None, None, None, None) == -1
    assert lib.rtr_host_build_bvh(None, None, None, 0, None, 0) == -1


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under realtimeraytracer_amd/ may import, link or include it."""
    pkg = os.path.join(ROOT, "realtimeraytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep) or "__pycache__" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                code = re.sub(r"/\*.*?\*/|//[^\n]*|#[^\n]*|\"\"\".*?\"\"\"", "", text, flags=re.S)
                assert "oracle_py" not in code and "liboracle" not in code and "oracle.h" not in code and "oracle/" not in code, \
                    f"{fn} references the oracle"
    ldd = subprocess.check_output(["ldd", A.LIB_HIP_PATH]).decode()
    assert "oracle" not in ldd


def test_shipped_libraries_hold_no_test_hooks():
    """environment switches that let ranks share a device, make a one-rank communicator exchange with itself, shrink the
    overflow list or force the visibility pre-fill must not be reachable in the shipped libraries.  They are compiled only into
    librtr_hip_test.so / librtr_mgpu_test.so (-DRTR_TEST_HOOKS, same sources), which export the same ABI."""
    hooks = (b"RTR_MGPU_TEST_SHARED_DEVICE", b"RTR_MGPU_SELF_EXCHANGE", b"RTR_TRACE_OVERFLOW_CAP", b"RTR_TRACE_VIS_FILL", b"RTR_MGPU_TEST_WRONG_PLACE")
    for path in (A.LIB_HIP_PATH, A.LIB_MGPU_PATH, A.LIB_HOST_PATH):
        blob = open(path, "rb").read()
        for h in hooks:
            assert h not in blob, f"{os.path.basename(path)} contains {h.decode()}"
    assert all(h in open(A.LIB_HIP_HOOKS_PATH, "rb").read() for h in hooks[2:4])
    assert all(h in open(A.LIB_MGPU_HOOKS_PATH, "rb").read() for h in hooks[:2] + hooks[4:])
    # the test builds export every entry point the headers declare (loading them needs no GPU)
    A.hip_lib_with_hooks()
    for name in A.MGPU_SYMBOLS:
        assert hasattr(C.CDLL(A.LIB_MGPU_HOOKS_PATH), name)
    # and the product reads no tunable per render: one place reads the environment, when a context is made
    src = open(os.path.join(ROOT, "realtimeraytracer_amd", "csrc", "kernels", "rtr_kernels.hip")).read()
    assert src.count("getenv(") == 1 and "tunables_from_env" in src


def test_scene_limits_at_the_boundary_of_the_32_bit_record_offsets():
    """the traversal kernels address records through 32-bit byte offsets, so a scene whose triangle records (48 B) or
    4-wide records (64 B) would reach 2 GiB is refused before anything is built — checked on the arithmetic, no giant allocation."""
    lib = A.hip_lib()
    max_tris = (2**31 - 1) // 48            # 44 739 242
    max_nodes = (2**31 - 1) // 64           # 33 554 431
    assert lib.rtr_check_scene_limits(max_tris, 0) == 0
    assert lib.rtr_check_scene_limits(max_tris + 1, 0) == -1 and b"triangle records would reach 2 GiB" in lib.rtr_last_error()
    assert lib.rtr_check_scene_limits(1000, max_nodes) == 0
    assert lib.rtr_check_scene_limits(1000, max_nodes + 1) == -1 and b"4-wide records would reach 2 GiB" in lib.rtr_last_error()
    assert lib.rtr_check_scene_limits(2**28, 0) == -1 and b"leaf encoding" in lib.rtr_last_error()
    # what rtr_scene_create asks: every triangle in a leaf of its own -> numTriangles - 1 nodes; the node limit binds first
    assert lib.rtr_check_scene_limits(max_nodes + 1, max_nodes) == 0
    assert lib.rtr_check_scene_limits(max_nodes + 2, max_nodes + 1) == -1
    assert lib.rtr_check_scene_limits(0, 0) == 0
