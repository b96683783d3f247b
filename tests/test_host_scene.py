"""Host scene layer (csrc/host/): scene::Camera, scene::Object, scene::AreaLight, CreateScene — method names,
argument meaning and quirks of the reference (src/scene/*.cppm)."""
import math

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import host, scenes


def test_camera_known_answer():
    # fovY=60, pos=(0,0,5), lookAt=0, up=+Y, 1600x1200 (the reference's constructor call, application.cppm:74-81)
    cam = host.Camera(60.0, (0, 0, 5), (0, 0, 0), (0, 1, 0), 1600, 1200)
    st = cam.state()
    assert abs(st["yaw"] + 90.0) < 1e-4 and abs(st["pitch"]) < 1e-4
    d = cam.getGPUData()
    np.testing.assert_allclose(d.horizontalViewportDelta[:], (9.6225052e-4, 0.0, 0.0), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(d.verticalViewportDelta[:], (0.0, -9.6225e-4, 0.0), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(d.topLeftViewportCorner[:], (-0.7698005, 0.5773503, 4.0), rtol=1e-6)
    assert tuple(d.position[:]) == (0.0, 0.0, 5.0)
    np.testing.assert_allclose(st["forward"], (0, 0, -1), atol=1e-6)
    np.testing.assert_allclose(st["right"], (1, 0, 0), atol=1e-6)


def test_camera_controls():
    cam = host.Camera(60.0, (0, 0, 5), (0, 0, 0), (0, 1, 0), 800, 600)
    cam.processMouseMovement(100.0, 1000.0)       # sensitivity 0.1, pitch clamped to 89 (camera.cppm:136-148)
    st = cam.state()
    assert abs(st["yaw"] - (-80.0)) < 1e-4 and st["pitch"] == 89.0
    cam.rotateY(0.1)                              # adds to the DEGREES yaw, as the reference does (camera.cppm:149-154)
    assert abs(cam.state()["yaw"] - (-79.9)) < 1e-4
    cam.setPosition((1, 2, 3))
    assert tuple(cam.getGPUData().position[:]) == (1.0, 2.0, 3.0)


def test_headless_input_known_answers():
    """csrc/host/input.hpp = Window::processInput + the mouse callback of the reference (window.cppm:68-133) on scripted input:
    W / S / A / D step by forward / right x CAM_SPEED (10.5, application.cppm:497), T toggles the 0.1-degree-per-frame spin on its
    press edge, cursor travel is scaled by MOUSE_SENSITIVITY (0.5) before the camera's own 0.1 degrees per unit."""
    cam = host.Camera(60.0, (0, 0, 5), (0, 0, 0), (0, 1, 0), 800, 600)
    assert cam.applyInput("W") is False
    np.testing.assert_allclose(cam.getGPUData().position[:], (0, 0, 5 - 10.5), atol=1e-5)
    cam.applyInput("D")
    np.testing.assert_allclose(cam.getGPUData().position[:], (10.5, 0, -5.5), atol=1e-5)
    cam.applyInput("SA")                                           # both held in one frame
    np.testing.assert_allclose(cam.getGPUData().position[:], (0, 0, 5), atol=1e-5)
    cam.applyInput("ws")                                           # opposite keys cancel; the position is still rewritten
    np.testing.assert_allclose(cam.getGPUData().position[:], (0, 0, 5), atol=1e-5)
    cam.applyInput("", mouse=(40.0, -20.0))                        # 40 * 0.5 * 0.1 = 2 degrees of yaw, -1 of pitch
    st = cam.state()
    assert abs(st["yaw"] - (-88.0)) < 1e-4 and abs(st["pitch"] - (-1.0)) < 1e-4
    assert cam.applyInput("T") is True                             # press edge: spin on, and this frame already turns
    assert abs(cam.state()["yaw"] - (-87.9)) < 1e-4
    assert cam.applyInput("T") is True                             # still held: no second toggle
    assert cam.applyInput("") is True                              # released: keeps spinning
    assert abs(cam.state()["yaw"] - (-87.7)) < 1e-4
    assert cam.applyInput("T") is False                            # second press: off
    assert abs(cam.state()["yaw"] - (-87.7)) < 1e-4
    cam.applyInput("W", cam_speed=2.0)
    f = np.array(cam.state()["forward"])
    np.testing.assert_allclose(cam.getGPUData().position[:], np.array((0, 0, 5.0)) + 2.0 * f, atol=1e-5)


def test_object_transform_quirks():
    hs = host.HostScene()
    o = hs.addObject("square")
    o.move((1, 2, 3)).scale(2.0)
    m = o.getTransform()
    np.testing.assert_array_equal(m, [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3]])   # scale leaves the translation alone
    o2 = hs.addObject("square")
    o2.rotate((0.0, 0.0, 90.0))
    r = o2.getTransform()[:, :3]
    # quirk Q4: rotation[row][k] indexes a column-major mat3 -> the TRANSPOSE of Rz(90) is applied
    np.testing.assert_allclose(r, [[0, 1, 0], [-1, 0, 0], [0, 0, 1]], atol=1e-6)


def test_area_light_scale_is_diagonal_only_and_packing():
    hs = host.HostScene()
    l = hs.addAreaLight(9.0, (0.8, 0.5, 0.2), False)
    l.move((-1600.0, 2500.5, -500.0)).scale((1300.0, 750.5, 100.0)).rotate((0.0, 90.0, 0.0)).rotate((0.0, 0.0, 55.0))  # application.cppm:184-188
    hs.build()
    li = hs.lightInfos()[0]
    t = l.getTransform()
    # PackTransformMatrix (core/utils.cppm:11-39): column-major mat4, last row 0 0 0 1
    m = np.array(li.transform[:], dtype=np.float32).reshape(4, 4).T
    np.testing.assert_array_equal(m[:3, :], t)
    np.testing.assert_array_equal(m[3], [0, 0, 0, 1])
    assert li.numTriangles == 2 and li.vertexOffset == 0 and li.indexOffset == 0 and li.isTwoSided == 0
    assert abs(li.intensity - 9.0) < 1e-6
    # quirk Q5: a later scale multiplies only the diagonal
    before = l.getTransform().copy()
    l.scale((2.0, 2.0, 2.0))
    after = l.getTransform()
    assert after[0, 0] == before[0, 0] * 2 and after[0, 1] == before[0, 1] and after[1, 0] == before[1, 0]


def test_square_light_geometry_and_instance_order(scene_cache):
    s = scenes.cornell_box(64, 64)
    hs = s.host
    inst = hs.instances()
    assert [i.customIndex for i in inst] == list(range(len(inst)))          # tlas.cppm:63-82
    assert inst[0].meshIndex == 0                                           # lights first
    v = hs.vertices()
    np.testing.assert_array_equal(v[:4, :3], [[-0.5, -0.5, 0], [-0.5, 0.5, 0], [0.5, 0.5, 0], [0.5, -0.5, 0]])   # area_light.cppm:79-82
    np.testing.assert_array_equal(hs.indices()[:6], [0, 1, 2, 0, 2, 3])     # geometry_builder.cppm:88
    assert hs.meshes()[0].isOpaque == 1 and hs.meshes()[1].isOpaque == 0    # quirk Q11: OBJ/MTL meshes are non-opaque


def test_model_dedup_across_instances(scene_cache):
    obj, _ = scenes.write_cornell(scene_cache)
    hs = host.HostScene()
    a = hs.addObject(obj)
    b = hs.addObject(obj)
    b.move((10, 0, 0))
    hs.build()
    assert hs.desc.numMeshes == 1 and hs.desc.numInstances == 2             # geometry_builder.cppm:67-76 path-keyed cache
    assert a.info()["blasIndex"] == b.info()["blasIndex"] == 0
    assert a.info()["numTriangles"] == 36
    # loadModel de-duplicates across the whole file: 36 tris over the shared floor/wall corners
    assert hs.desc.numIndices == 108


def test_missing_textures_are_refused_loudly(scene_cache):
    """A texture the host cannot load, or an ObjectInfo that names a texture the caller did not supply, is an error —
    the library never substitutes a constant."""
    import ctypes as C
    obj, _ = scenes.write_cornell(scene_cache)
    hs = host.HostScene()
    o = hs.addObject(obj)
    o.setColor("some/missing_texture.png")
    with pytest.raises(host.HostError):                  # createTextureImage: "Image load failed" (file.cppm:282-286)
        hs.build()
    s = scenes.cornell_box(32, 32)
    d = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    objs = (A.RtrObjectInfo * d.numObjects)(*[s.desc.objects[i] for i in range(d.numObjects)])
    objs[2].usesColorMap, objs[2].colorIndex = 1, 7
    d.objects = C.cast(objs, C.POINTER(A.RtrObjectInfo))
    lib = A.hip_lib()
    st = A.rtr_scene_stats()
    rc = lib.rtr_host_build_bvh(C.byref(d), C.byref(st), None, 0, None, 0)
    assert rc == -1 and b"texture index 7" in lib.rtr_last_error()


def test_reference_ltc_tables_known_answers_and_analytic_image(oracle, scene_cache):
    """Where the reference tree exists: its real LTC tables (external/LUT/ltc_matrix.h, dumped by oracle/_ref/ltc_dump —
    the data itself is not committed) satisfy SURVEY Appendix B and drive the oracle's analytic (LTC) image to a finite,
    non-trivial result with BVH == brute force.  Elsewhere the synthetic stand-ins of scenes.synthetic_ltc() are used."""
    import os
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ltc_dump")
    if not os.path.exists(tool):
        pytest.skip("reference tree absent: oracle/_ref/ltc_dump not built")
    a = np.frombuffer(subprocess.run([tool], check=True, capture_output=True).stdout, np.float32)
    assert a.size == 2 * 16384
    l1, l2 = a[:16384].reshape(64, 64, 4).copy(), a[16384:].reshape(64, 64, 4).copy()
    assert l1.reshape(-1)[3] == np.float32(2e-05) and l2.reshape(-1)[:4].tolist() == [1.0, 0.0, 0.0, 0.0]
    assert abs(float(l1.astype(np.float64).sum()) - 5409.868614) < 1e-5 and abs(float(l2.astype(np.float64).sum()) - 4783.065395) < 1e-5
    from realtimeraytracer_amd import api
    s = scenes.cornell_box(96, 96, ltc=(l1, l2))
    p = api.make_params(96, 96, spp=1, images=A.IMAGES_RAYGEN5)
    st, nodes, tris = api.host_build_bvh(s.desc)
    r = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=(nodes, tris, st.grid), images=A.IMAGES_RAYGEN5, threads=8)
    b = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=None, images=A.IMAGES_RAYGEN5, threads=8)
    an = r.images[A.IMAGE_ANALYTIC].view(np.uint8).reshape(96, 96, 4)
    assert np.array_equal(r.images[A.IMAGE_ANALYTIC], b.images[A.IMAGE_ANALYTIC])
    assert np.all(an[..., 3] == 255) and an[..., :3].std() > 10 and an[..., :3].max() > 100


def test_desc_with_inconsistent_object_offsets_is_refused(scene_cache):
    """The hit shader fetches through ObjectInfo.vertexOffset / indexOffset: a desc whose ObjectInfo disagrees with the mesh of
    its instance would read another mesh's data (or past the arrays) on the device, so scene creation refuses it."""
    import ctypes as C
    from realtimeraytracer_amd import api
    s = scenes.cornell_box(32, 32)
    api.host_build_bvh(s.desc)                                        # the honest desc is accepted
    d = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    objs = (A.RtrObjectInfo * d.numObjects)(*[A.RtrObjectInfo.from_buffer_copy(bytes(d.objects[i])) for i in range(d.numObjects)])
    objs[1].vertexOffset += 3
    d.objects = C.cast(objs, C.POINTER(A.RtrObjectInfo))
    with pytest.raises(api.RtrError):
        api.host_build_bvh(d)
    objs[1].vertexOffset -= 3
    objs[2].indexOffset = 10 ** 9
    with pytest.raises(api.RtrError):
        api.host_build_bvh(d)
