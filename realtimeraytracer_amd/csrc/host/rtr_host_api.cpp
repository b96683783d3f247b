// rtr_host_api.cpp — C entry points over the C++ host scene layer (scene.hpp, scene_builder.hpp) so the
// Python tests / bench can drive scene::Camera, scene::Object, scene::AreaLight, core::file ingest and
// app::setup::CreateScene exactly as reference src/app/application.cppm:181-230 does, then hand the
// packed arrays to the C ABI (rtr_scene_create) and to the CPU oracle.  Built as librtr_host.so.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "input.hpp"
#include "scene_builder.hpp"

namespace {
thread_local std::string g_herr;
template <class F>
int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_herr = e.what(); return -1; }
    catch (...) { g_herr = "unknown exception"; return -1; }
}
}  // namespace

struct rtrh_scene {
    std::vector<std::shared_ptr<scene::Object>> objects;
    std::vector<std::shared_ptr<scene::AreaLight>> lights;
    std::vector<std::pair<std::string, std::string>> objMtlPairs;
    app::setup::SceneReturnInfo built;
    bool isBuilt = false;
    std::vector<float> ltc1, ltc2;
    rtr::vm::vec3 sky{0.f, 0.f, 0.f};
    std::string hdriPath;
};

struct rtrh_camera { std::unique_ptr<scene::Camera> cam; };

extern "C" {

const char* rtrh_last_error(void) { return g_herr.c_str(); }

rtrh_scene* rtrh_scene_new(void) { return new rtrh_scene(); }
void rtrh_scene_free(rtrh_scene* s) { delete s; }

int rtrh_add_light(rtrh_scene* s, float intensity, const float* color, int twoSided, int visible, const char* objPath) {
    int idx = -1;
    int rc = guarded([&] {
        rtr::vm::vec3 c(color[0], color[1], color[2]);
        auto l = objPath && *objPath ? std::make_shared<scene::AreaLight>(intensity, c, twoSided != 0, visible != 0, std::string(objPath))
                                     : std::make_shared<scene::AreaLight>(intensity, c, twoSided != 0, visible != 0);
        s->lights.push_back(l);
        idx = (int)s->lights.size() - 1;
    });
    return rc ? rc : idx;
}
int rtrh_light_move(rtrh_scene* s, int i, const float* v) { return guarded([&] { s->lights.at(i)->move({v[0], v[1], v[2]}); }); }
int rtrh_light_scale(rtrh_scene* s, int i, const float* v) { return guarded([&] { s->lights.at(i)->scale({v[0], v[1], v[2]}); }); }
int rtrh_light_rotate(rtrh_scene* s, int i, const float* v) { return guarded([&] { s->lights.at(i)->rotate({v[0], v[1], v[2]}); }); }
int rtrh_light_transform(rtrh_scene* s, int i, float* out12) {
    return guarded([&] { auto t = s->lights.at(i)->getTransform(); std::memcpy(out12, t.matrix, 12 * sizeof(float)); });
}

int rtrh_add_object(rtrh_scene* s, const char* objPath) {
    int idx = -1;
    int rc = guarded([&] { s->objects.push_back(std::make_shared<scene::Object>(std::string(objPath))); idx = (int)s->objects.size() - 1; });
    return rc ? rc : idx;
}
int rtrh_object_move(rtrh_scene* s, int i, const float* v) { return guarded([&] { s->objects.at(i)->move({v[0], v[1], v[2]}); }); }
int rtrh_object_scale(rtrh_scene* s, int i, float k) { return guarded([&] { s->objects.at(i)->scale(k); }); }
int rtrh_object_rotate(rtrh_scene* s, int i, const float* v) { return guarded([&] { s->objects.at(i)->rotate({v[0], v[1], v[2]}); }); }
int rtrh_object_set_color(rtrh_scene* s, int i, const float* c) { return guarded([&] { s->objects.at(i)->setColor(rtr::vm::vec3(c[0], c[1], c[2])); }); }
int rtrh_object_set_color_map(rtrh_scene* s, int i, const char* p) { return guarded([&] { s->objects.at(i)->setColor(std::string(p)); }); }
int rtrh_object_set_specular(rtrh_scene* s, int i, float v) { return guarded([&] { s->objects.at(i)->setSpecular(v); }); }
int rtrh_object_set_metallic(rtrh_scene* s, int i, float v) { return guarded([&] { s->objects.at(i)->setMetallic(v); }); }
int rtrh_object_transform(rtrh_scene* s, int i, float* out12) {
    return guarded([&] { auto t = s->objects.at(i)->getTransform(); std::memcpy(out12, t.matrix, 12 * sizeof(float)); });
}
int rtrh_num_objects(rtrh_scene* s) { return (int)s->objects.size(); }
int rtrh_num_lights(rtrh_scene* s) { return (int)s->lights.size(); }

int rtrh_add_obj_mtl_pair(rtrh_scene* s, const char* objPath, const char* mtlDir) {
    return guarded([&] { s->objMtlPairs.emplace_back(std::string(objPath), std::string(mtlDir ? mtlDir : "")); });
}
int rtrh_set_ltc(rtrh_scene* s, const float* ltc1, const float* ltc2) {
    return guarded([&] { s->ltc1.assign(ltc1, ltc1 + 64 * 64 * 4); s->ltc2.assign(ltc2, ltc2 + 64 * 64 * 4); });
}
int rtrh_set_hdri(rtrh_scene* s, const char* path) { return guarded([&] { s->hdriPath = path ? path : ""; }); }
// decode an image exactly as core::file::createTextureImage does (flip + channel conversion); out may be NULL to query the extent
int rtrh_load_image(const char* path, int grayscale, int* w, int* h, uint8_t* out, size_t outBytes) {
    return guarded([&] {
        rtr::img::Image im = core::file::createTextureImage(path, grayscale != 0);
        if (w) *w = im.width;
        if (h) *h = im.height;
        if (out) {
            if (outBytes != im.pixels.size()) throw std::runtime_error("rtrh_load_image: buffer size mismatch");
            std::memcpy(out, im.pixels.data(), outBytes);
        }
    });
}
int rtrh_set_sky(rtrh_scene* s, const float* c) { return guarded([&] { s->sky = rtr::vm::vec3(c[0], c[1], c[2]); }); }

// runs CreateScene::createSceneFromObjectsAndLights (reference application.cppm:230)
int rtrh_build(rtrh_scene* s) {
    return guarded([&] {
        s->built = app::setup::CreateScene::createSceneFromObjectsAndLights(s->objects, s->objMtlPairs, s->lights);
        if (!s->hdriPath.empty()) s->built.setHDRI(s->hdriPath);
        s->objMtlPairs.clear();   // their shapes are Objects now; a second build must not ingest them again
        s->isBuilt = true;
    });
}
// descriptor for rtr_scene_create / the oracle; pointers are owned by `s` and valid until the next build/free
int rtrh_get_desc(rtrh_scene* s, rtr_scene_desc* out) {
    return guarded([&] {
        if (!s->isBuilt) throw std::runtime_error("rtrh_get_desc: call rtrh_build first");
        *out = s->built.desc(s->ltc1.empty() ? nullptr : s->ltc1.data(), s->ltc2.empty() ? nullptr : s->ltc2.data(), s->sky);
    });
}
int rtrh_object_info(rtrh_scene* s, int i, uint32_t* blasIndex, uint32_t* instanceIndex, uint32_t* numTriangles) {
    return guarded([&] {
        auto& o = s->objects.at(i);
        if (blasIndex) *blasIndex = o->getBLASIndex();
        if (instanceIndex) *instanceIndex = o->getInstanceIndex();
        if (numTriangles) *numTriangles = o->getNumTriangles();
    });
}

// standalone ingest entry points (file.cppm:44-102 and :112-269) for the OBJ parity tests:
// results are returned through a fresh rtrh_scene whose desc holds the arrays
int rtrh_load_model(rtrh_scene* s, const char* path) {
    return guarded([&] {
        s->built = app::setup::SceneReturnInfo();
        auto& g = s->built.geoReturnInfo;
        core::file::loadModel(path, g.vertexPositions, g.indices, g.vertices);
        RtrMesh m{}; m.vertexCount = (uint32_t)g.vertices.size(); m.indexCount = (uint32_t)g.indices.size(); m.isOpaque = 1;
        g.meshes.push_back(m);
        s->isBuilt = true;
    });
}

// Dump the raw OBJ reader output (rtr::obj::LoadObj) in EXACTLY the text format of
// oracle/ref_tinyobj_dump.cpp, so the two files can be compared byte for byte.
int rtrh_obj_dump(const char* objPath, const char* mtlDir, const char* outPath) {
    return guarded([&] {
        rtr::obj::attrib_t attrib; std::vector<rtr::obj::shape_t> shapes; std::vector<rtr::obj::material_t> materials;
        std::string warn, err;
        const char* mtl = mtlDir && *mtlDir ? mtlDir : nullptr;
        if (!rtr::obj::LoadObj(&attrib, &shapes, &materials, &warn, &err, objPath, mtl)) throw std::runtime_error(warn + err);
        FILE* f = std::fopen(outPath, "w");
        if (!f) throw std::runtime_error(std::string("cannot open ") + outPath);
        auto farr = [&](const char* name, const std::vector<float>& v) {
            std::fprintf(f, "\"%s\":[", name);
            for (size_t i = 0; i < v.size(); ++i) std::fprintf(f, "%s%.9g", i ? "," : "", v[i]);
            std::fprintf(f, "],");
        };
        std::fprintf(f, "{");
        farr("vertices", attrib.vertices); farr("normals", attrib.normals); farr("texcoords", attrib.texcoords);
        std::fprintf(f, "\"shapes\":[");
        for (size_t s2 = 0; s2 < shapes.size(); ++s2) {
            const auto& m = shapes[s2].mesh;
            std::fprintf(f, "%s{\"name\":\"%s\",\"indices\":[", s2 ? "," : "", shapes[s2].name.c_str());
            for (size_t i = 0; i < m.indices.size(); ++i)
                std::fprintf(f, "%s[%d,%d,%d]", i ? "," : "", m.indices[i].vertex_index, m.indices[i].normal_index, m.indices[i].texcoord_index);
            std::fprintf(f, "],\"num_face_vertices\":[");
            for (size_t i = 0; i < m.num_face_vertices.size(); ++i) std::fprintf(f, "%s%u", i ? "," : "", (unsigned)m.num_face_vertices[i]);
            std::fprintf(f, "],\"material_ids\":[");
            for (size_t i = 0; i < m.material_ids.size(); ++i) std::fprintf(f, "%s%d", i ? "," : "", m.material_ids[i]);
            std::fprintf(f, "]}");
        }
        std::fprintf(f, "],\"materials\":[");
        for (size_t i = 0; i < materials.size(); ++i) {
            const auto& m = materials[i];
            std::fprintf(f, "%s{\"name\":\"%s\",\"diffuse\":[%.9g,%.9g,%.9g],\"specular\":[%.9g,%.9g,%.9g],", i ? "," : "", m.name.c_str(),
                         m.diffuse[0], m.diffuse[1], m.diffuse[2], m.specular[0], m.specular[1], m.specular[2]);
            std::fprintf(f, "\"diffuse_texname\":\"%s\",\"specular_texname\":\"%s\",\"metallic_texname\":\"%s\",\"alpha_texname\":\"%s\",\"unknown\":{",
                         m.diffuse_texname.c_str(), m.specular_texname.c_str(), m.metallic_texname.c_str(), m.alpha_texname.c_str());
            size_t k = 0;
            for (const auto& kv : m.unknown_parameter) std::fprintf(f, "%s\"%s\":\"%s\"", k++ ? "," : "", kv.first.c_str(), kv.second.c_str());
            std::fprintf(f, "}}");
        }
        std::fprintf(f, "]}\n");
        std::fclose(f);
    });
}

/* ---- camera ------------------------------------------------------------------------------------ */
rtrh_camera* rtrh_camera_new(float fovY, const float* pos, const float* lookAt, const float* up, int w, int h) {
    rtrh_camera* c = nullptr;
    guarded([&] {
        c = new rtrh_camera();
        c->cam = std::make_unique<scene::Camera>(fovY, rtr::vm::vec3(pos[0], pos[1], pos[2]), rtr::vm::vec3(lookAt[0], lookAt[1], lookAt[2]),
                                                 rtr::vm::vec3(up[0], up[1], up[2]), w, h);
    });
    return c;
}
void rtrh_camera_free(rtrh_camera* c) { delete c; }
// one frame of the headless input path (app::applyInput); state[0] = spinning, state[1] = T was down (in / out)
int rtrh_camera_apply_input(rtrh_camera* c, const char* keys, float mouseDx, float mouseDy, float camSpeed, float mouseSensitivity, int* state) {
    return guarded([&] {
        bool spinning = state[0] != 0, tWasDown = state[1] != 0;
        app::applyInput(*c->cam, keys ? keys : "", mouseDx, mouseDy, camSpeed, mouseSensitivity, spinning, tWasDown);
        state[0] = spinning ? 1 : 0; state[1] = tWasDown ? 1 : 0;
    });
}
int rtrh_camera_get(rtrh_camera* c, RtrCameraData* out) { return guarded([&] { *out = c->cam->getGPUData(); }); }
int rtrh_camera_set_position(rtrh_camera* c, const float* p) { return guarded([&] { c->cam->setPosition({p[0], p[1], p[2]}); }); }
int rtrh_camera_rotate_y(rtrh_camera* c, float a) { return guarded([&] { c->cam->rotateY(a); }); }
int rtrh_camera_mouse(rtrh_camera* c, float dx, float dy) { return guarded([&] { c->cam->processMouseMovement(dx, dy); }); }
int rtrh_camera_state(rtrh_camera* c, float* out8) {   // yaw, pitch, forward xyz, right xyz
    return guarded([&] {
        out8[0] = c->cam->getYaw(); out8[1] = c->cam->getPitch();
        auto f = c->cam->getForward(); auto r = c->cam->getRight();
        out8[2] = f.x; out8[3] = f.y; out8[4] = f.z; out8[5] = r.x; out8[6] = r.y; out8[7] = r.z;
    });
}

}  // extern "C"
