"""Python handles over librtr_host.so — the C shim of the C++ host scene layer (csrc/host/):
scene::Camera, scene::Object, scene::AreaLight, core::file ingest, app::setup::CreateScene.
Method names follow the reference classes (src/scene/*.cppm) so tests read like reference usage
(src/app/application.cppm:181-230)."""
import ctypes as C

import numpy as np

from . import _abi as A


class HostError(RuntimeError):
    pass


def _v3(v):
    return (A.f32 * 3)(*[float(x) for x in v])


def _chk(rc, what):
    if rc < 0:
        raise HostError(f"{what}: {A.host_lib().rtrh_last_error().decode()}")
    return rc


class Camera:
    """scene::Camera (reference src/scene/camera.cppm:19-154) without the Vulkan device/buffer."""

    def __init__(self, fovY, position, lookAt, up, pixelWidth, pixelHeight):
        self.lib = A.host_lib()
        self.h = self.lib.rtrh_camera_new(float(fovY), _v3(position), _v3(lookAt), _v3(up), int(pixelWidth), int(pixelHeight))
        if not self.h:
            raise HostError("rtrh_camera_new failed: " + self.lib.rtrh_last_error().decode())

    def getGPUData(self):
        d = A.RtrCameraData()
        _chk(self.lib.rtrh_camera_get(self.h, C.byref(d)), "getGPUData")
        return d

    def setPosition(self, p):
        _chk(self.lib.rtrh_camera_set_position(self.h, _v3(p)), "setPosition")

    def rotateY(self, angle):
        _chk(self.lib.rtrh_camera_rotate_y(self.h, float(angle)), "rotateY")

    def processMouseMovement(self, dx, dy):
        _chk(self.lib.rtrh_camera_mouse(self.h, float(dx), float(dy)), "processMouseMovement")

    def applyInput(self, keys="", mouse=(0.0, 0.0), cam_speed=10.5, mouse_sensitivity=0.5):
        """one frame of the headless input path (csrc/host/input.hpp = Window::processInput + mouse callback of the reference)"""
        if not hasattr(self, "_input_state"):
            self._input_state = (C.c_int * 2)(0, 0)
        _chk(self.lib.rtrh_camera_apply_input(self.h, keys.encode(), float(mouse[0]), float(mouse[1]), float(cam_speed), float(mouse_sensitivity),
                                              self._input_state), "applyInput")
        return bool(self._input_state[0])

    def state(self):
        out = (A.f32 * 8)()
        _chk(self.lib.rtrh_camera_state(self.h, out), "state")
        return {"yaw": out[0], "pitch": out[1], "forward": tuple(out[2:5]), "right": tuple(out[5:8])}

    def __del__(self):
        try:
            if self.h:
                self.lib.rtrh_camera_free(self.h)
                self.h = None
        except Exception:
            pass


class _Light:
    def __init__(self, scene, idx):
        self.s, self.i = scene, idx

    def move(self, v):
        _chk(self.s.lib.rtrh_light_move(self.s.h, self.i, _v3(v)), "AreaLight.move")
        return self

    def scale(self, v):
        _chk(self.s.lib.rtrh_light_scale(self.s.h, self.i, _v3(v)), "AreaLight.scale")
        return self

    def rotate(self, deg):
        _chk(self.s.lib.rtrh_light_rotate(self.s.h, self.i, _v3(deg)), "AreaLight.rotate")
        return self

    def getTransform(self):
        out = (A.f32 * 12)()
        _chk(self.s.lib.rtrh_light_transform(self.s.h, self.i, out), "AreaLight.getTransform")
        return np.array(out, dtype=np.float32).reshape(3, 4)


class _Object:
    def __init__(self, scene, idx):
        self.s, self.i = scene, idx

    def move(self, v):
        _chk(self.s.lib.rtrh_object_move(self.s.h, self.i, _v3(v)), "Object.move")
        return self

    def scale(self, k):
        _chk(self.s.lib.rtrh_object_scale(self.s.h, self.i, float(k)), "Object.scale")
        return self

    def rotate(self, deg):
        _chk(self.s.lib.rtrh_object_rotate(self.s.h, self.i, _v3(deg)), "Object.rotate")
        return self

    def setColor(self, c):
        if isinstance(c, str):
            _chk(self.s.lib.rtrh_object_set_color_map(self.s.h, self.i, c.encode()), "Object.setColor(path)")
        else:
            _chk(self.s.lib.rtrh_object_set_color(self.s.h, self.i, _v3(c)), "Object.setColor")
        return self

    def setSpecular(self, v):
        _chk(self.s.lib.rtrh_object_set_specular(self.s.h, self.i, float(v)), "Object.setSpecular")
        return self

    def setMetallic(self, v):
        _chk(self.s.lib.rtrh_object_set_metallic(self.s.h, self.i, float(v)), "Object.setMetallic")
        return self

    def getTransform(self):
        out = (A.f32 * 12)()
        _chk(self.s.lib.rtrh_object_transform(self.s.h, self.i, out), "Object.getTransform")
        return np.array(out, dtype=np.float32).reshape(3, 4)

    def info(self):
        b, i, n = A.u32(), A.u32(), A.u32()
        _chk(self.s.lib.rtrh_object_info(self.s.h, self.i, C.byref(b), C.byref(i), C.byref(n)), "Object.info")
        return {"blasIndex": b.value, "instanceIndex": i.value, "numTriangles": n.value}


class HostScene:
    """What Application::run() assembles before creating GPU state (reference application.cppm:181-230):
    lights, explicit objects, OBJ/MTL pairs -> CreateScene::createSceneFromObjectsAndLights."""

    def __init__(self):
        self.lib = A.host_lib()
        self.h = self.lib.rtrh_scene_new()
        self.desc = None
        self._keep = []

    def addAreaLight(self, intensity, color, isTwoSided=False, isVisible=True, objPath=None):
        i = _chk(self.lib.rtrh_add_light(self.h, float(intensity), _v3(color), int(isTwoSided), int(isVisible),
                                         objPath.encode() if objPath else None), "addAreaLight")
        return _Light(self, i)

    def addObject(self, objPath):
        i = _chk(self.lib.rtrh_add_object(self.h, objPath.encode()), "addObject")
        return _Object(self, i)

    def object(self, i):
        return _Object(self, i)

    def addObjMtlPair(self, objPath, mtlDir=""):
        _chk(self.lib.rtrh_add_obj_mtl_pair(self.h, objPath.encode(), mtlDir.encode()), "addObjMtlPair")

    def setLTC(self, ltc1, ltc2):
        a = np.ascontiguousarray(ltc1, dtype=np.float32).ravel()
        b = np.ascontiguousarray(ltc2, dtype=np.float32).ravel()
        assert a.size == 64 * 64 * 4 and b.size == 64 * 64 * 4
        _chk(self.lib.rtrh_set_ltc(self.h, a.ctypes.data_as(C.POINTER(A.f32)), b.ctypes.data_as(C.POINTER(A.f32))), "setLTC")

    def setHDRI(self, path):
        """equirect sky image (reference application.cppm:250: createTextureImage(sky4k.hdr, false))"""
        _chk(self.lib.rtrh_set_hdri(self.h, path.encode() if path else None), "setHDRI")

    def setSky(self, c):
        _chk(self.lib.rtrh_set_sky(self.h, _v3(c)), "setSky")

    def build(self):
        _chk(self.lib.rtrh_build(self.h), "createSceneFromObjectsAndLights")
        d = A.rtr_scene_desc()
        _chk(self.lib.rtrh_get_desc(self.h, C.byref(d)), "get_desc")
        self.desc = d
        return d

    def loadModel(self, path):
        """core::file::loadModel (reference file.cppm:44-102): whole file -> one de-duplicated mesh."""
        _chk(self.lib.rtrh_load_model(self.h, path.encode()), "loadModel")
        d = A.rtr_scene_desc()
        _chk(self.lib.rtrh_get_desc(self.h, C.byref(d)), "get_desc")
        self.desc = d
        return d

    def numObjects(self):
        return self.lib.rtrh_num_objects(self.h)

    def numLights(self):
        return self.lib.rtrh_num_lights(self.h)

    # numpy views of the packed arrays (copies)
    def vertices(self):
        n = self.desc.numVertices
        return np.ctypeslib.as_array(C.cast(self.desc.vertices, C.POINTER(A.f32)), shape=(n, 12)).copy() if n else np.zeros((0, 12), np.float32)

    def indices(self):
        n = self.desc.numIndices
        return np.ctypeslib.as_array(self.desc.indices, shape=(n,)).copy() if n else np.zeros((0,), np.uint32)

    def meshes(self):
        return [self.desc.meshes[i] for i in range(self.desc.numMeshes)]

    def instances(self):
        return [self.desc.instances[i] for i in range(self.desc.numInstances)]

    def objectInfos(self):
        return [self.desc.objects[i] for i in range(self.desc.numObjects)]

    def lightInfos(self):
        return [self.desc.lights[i] for i in range(self.desc.numLights)]

    def __del__(self):
        try:
            if self.h:
                self.lib.rtrh_scene_free(self.h)
                self.h = None
        except Exception:
            pass


def load_image(path, grayscale=False):
    """core::file::createTextureImage's decode (reference file.cppm:272-291): flipped, RGBA8 or R8 numpy array."""
    lib = A.host_lib()
    w, h = C.c_int(), C.c_int()
    _chk(lib.rtrh_load_image(path.encode(), int(grayscale), C.byref(w), C.byref(h), None, 0), "load_image")
    ch = 1 if grayscale else 4
    out = np.empty((h.value, w.value, ch), np.uint8)
    _chk(lib.rtrh_load_image(path.encode(), int(grayscale), None, None, out.ctypes.data_as(A.VP), out.nbytes), "load_image")
    return out


def scene_info(frame, num_lights, cam_position):
    """scene::SceneInfo(frame, numAreaLights, camPosition) (reference scene_info.cppm:10-20)."""
    s = A.RtrSceneInfo()
    s.frame, s.numAreaLights = int(frame), int(num_lights)
    s.camPosition[0], s.camPosition[1], s.camPosition[2] = [float(x) for x in cam_position]
    return s
