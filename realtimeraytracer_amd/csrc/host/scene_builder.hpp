// scene_builder.hpp — mesh ingest and scene packing above the C ABI.
//
//   core::file::loadModel / loadOBJandMTL      <- reference src/core/file.cppm:44-269
//   app::setup::GeometryBuilder                <- reference src/app/setup/geometry_builder.cppm:50-212
//        (the array-packing half; the BLAS/TLAS half is the library's own BVH builder)
//   app::setup::CreateScene                    <- reference src/app/setup/create_scene.cppm:48-160
//        (GPUObjectInfo[] / GPUAreaLightInfo[] emission; texture upload is a "next" row)
//   instance ordering / customIndex / offset write-back  <- reference src/vulkan/raytracing/tlas.cppm:52-82
#pragma once
#include <algorithm>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/rtr.h"
#include "image_io.hpp"
#include "obj_loader.hpp"
#include "scene.hpp"

namespace core::file {

namespace vm = rtr::vm;
using scene::geometry::Vertex;

// What the reference keeps per BLAS (src/vulkan/raytracing/blas.cppm BLASCreateInfo) minus device addresses.
struct MeshData {
    uint32_t vertexIndexOffset = 0, indexIndexOffset = 0, vertexCount = 0, indexCount = 0;
    bool isOpaque = false;   // value-initialised false for OBJ/MTL meshes (file.cppm:254, quirk Q11)
};

// ---- OBJ ingest ------------------------------------------------------------------------------------------------------------
// Behaviour kept from the reference (src/core/file.cppm:44-269), pinned by tests/golden/*_tinyobj.json and cornell_ref_ingest.json:
//   * a corner (position / normal / uv triple of a face) becomes a 48-B Vertex; missing normals are zero, missing uvs are (0,0);
//   * equal Vertices are merged and indices are local to the mesh they belong to.  The two entry points differ only in how far
//     the merging reaches: loadModel merges across the whole FILE (one mesh), loadOBJandMTL within each SHAPE (one mesh + one
//     scene::Object per shape);
//   * a shape takes the material of its FIRST face (quirk Q12); without one it is grey, specular 1, metallic 0.
// The code below is organised as one corner walk feeding a de-duplicating pool whose lifetime is the merge scope.

// Resolves one OBJ corner.  The reference indexes the attribute arrays unchecked; a corner that names an attribute the file does
// not define is refused here instead of being read out of bounds.
inline Vertex corner_vertex(const rtr::obj::attrib_t& a, const rtr::obj::index_t& c) {
    const auto defined = [](int i, size_t width, const std::vector<float>& arr) { return i >= 0 && width * (size_t)i + (width - 1) < arr.size(); };
    if (!defined(c.vertex_index, 3, a.vertices) || (c.normal_index >= 0 && !defined(c.normal_index, 3, a.normals)) ||
        (c.texcoord_index >= 0 && !defined(c.texcoord_index, 2, a.texcoords)))
        throw std::runtime_error("OBJ: a face references a vertex, normal or texture coordinate that is not defined");
    Vertex out{};
    const float* p = &a.vertices[3 * (size_t)c.vertex_index];
    out.position = vm::vec3(p[0], p[1], p[2]);
    if (c.normal_index >= 0) { const float* n = &a.normals[3 * (size_t)c.normal_index]; out.normal = vm::vec3(n[0], n[1], n[2]); }
    if (c.texcoord_index >= 0) { const float* t = &a.texcoords[2 * (size_t)c.texcoord_index]; out.uv.x = t[0]; out.uv.y = t[1]; }
    return out;
}

// Vertices in first-seen order with mesh-local numbering; equal vertices share a number.
class VertexPool {
public:
    uint32_t intern(const Vertex& v) {
        const auto found = slot_.find(v);
        if (found != slot_.end()) return found->second;
        const uint32_t id = (uint32_t)ordered_.size();
        slot_.emplace(v, id);
        ordered_.push_back(v);
        return id;
    }
    const std::vector<Vertex>& ordered() const { return ordered_; }
private:
    std::unordered_map<Vertex, uint32_t> slot_;
    std::vector<Vertex> ordered_;
};

struct IngestedMesh {
    std::vector<Vertex> vertices;     // first-seen order
    std::vector<uint32_t> indices;    // local to `vertices`
    int firstFaceMaterial = -1;
};

enum class MergeScope { WholeFile, PerShape };

// The corner walk.  WholeFile: every corner of every shape, in file order, into one mesh (faces are taken as the loader
// delivers them).  PerShape: one mesh per shape; a face that is not a triangle is reported and skipped (the loader
// triangulates, so this only guards a loader that was told not to).
inline std::vector<IngestedMesh> ingest(const rtr::obj::attrib_t& attrib, const std::vector<rtr::obj::shape_t>& shapes, MergeScope scope) {
    std::vector<IngestedMesh> meshes;
    if (scope == MergeScope::WholeFile) {
        VertexPool pool;
        IngestedMesh whole;
        for (const rtr::obj::shape_t& sh : shapes)
            for (const rtr::obj::index_t& corner : sh.mesh.indices) whole.indices.push_back(pool.intern(corner_vertex(attrib, corner)));
        whole.vertices = pool.ordered();
        meshes.push_back(std::move(whole));
        return meshes;
    }
    for (const rtr::obj::shape_t& sh : shapes) {
        VertexPool pool;
        IngestedMesh m;
        m.firstFaceMaterial = sh.mesh.material_ids.empty() ? -1 : sh.mesh.material_ids.front();
        size_t corner = 0;
        for (const unsigned int arity : sh.mesh.num_face_vertices) {
            if (arity == 3) {
                for (size_t k = 0; k < 3; ++k) m.indices.push_back(pool.intern(corner_vertex(attrib, sh.mesh.indices[corner + k])));
            } else {
                std::cerr << "Non-triangle face detected, skipping face." << std::endl;
            }
            corner += arity;
        }
        m.vertices = pool.ordered();
        meshes.push_back(std::move(m));
    }
    return meshes;
}

// Appends a mesh to the scene-wide arrays and says where it landed.
inline MeshData append_mesh(const IngestedMesh& m, std::vector<vm::vec3>& vertexPositions, std::vector<uint32_t>& indices, std::vector<Vertex>& vertices) {
    MeshData where{};
    where.vertexIndexOffset = (uint32_t)vertices.size();
    where.indexIndexOffset = (uint32_t)indices.size();
    where.vertexCount = (uint32_t)m.vertices.size();
    where.indexCount = (uint32_t)m.indices.size();
    vertices.insert(vertices.end(), m.vertices.begin(), m.vertices.end());
    for (const Vertex& v : m.vertices) vertexPositions.push_back(v.position);
    indices.insert(indices.end(), m.indices.begin(), m.indices.end());
    return where;
}

inline std::string normalizePath(std::string path) {          // MTL files written on Windows name textures with back-slashes
    for (char& ch : path) if (ch == '\\') ch = '/';
    return path;
}

// MTL -> Object: a map wins over the constant of the same channel; texture paths are relative to the MTL directory.
inline void apply_material(scene::Object& obj, const rtr::obj::material_t* mat, const std::string& mtlDir) {
    if (!mat) { obj.setColor(vm::vec3(0.5f)); obj.setSpecular(1.0f); obj.setMetallic(0.0f); return; }
    const auto in_dir = [&](const std::string& name) { return normalizePath(mtlDir + name); };
    if (mat->diffuse_texname.empty()) obj.setColor(vm::vec3(mat->diffuse[0], mat->diffuse[1], mat->diffuse[2]));
    else obj.setColor(in_dir(mat->diffuse_texname));
    if (mat->specular_texname.empty()) obj.setSpecular(mat->specular[0]);
    else obj.setSpecular(in_dir(mat->specular_texname));
    const auto metallicParam = mat->unknown_parameter.find("metallic");
    if (!mat->metallic_texname.empty()) obj.setMetallic(in_dir(mat->metallic_texname));
    else if (metallicParam != mat->unknown_parameter.end()) obj.setMetallic(std::stof(metallicParam->second));
    else obj.setMetallic(0.0f);
    if (!mat->alpha_texname.empty()) obj.setOpacity(in_dir(mat->alpha_texname));
}

struct ParsedObj {
    std::vector<rtr::obj::material_t> materials;
    std::vector<rtr::obj::shape_t> shapes;
    rtr::obj::attrib_t attrib;
};
inline ParsedObj parse_obj(const std::string& objPath, const char* mtlDirOrNull) {
    ParsedObj parsed;
    std::string warnings, errors;
    const bool ok = rtr::obj::LoadObj(&parsed.attrib, &parsed.shapes, &parsed.materials, &warnings, &errors, objPath.c_str(), mtlDirOrNull);
    if (!ok) throw std::runtime_error(warnings + errors);
    return parsed;
}

// core::file::loadModel: the whole file as ONE mesh appended to the arrays; indices are local to that mesh.
inline void loadModel(const std::string& modelPath, std::vector<vm::vec3>& vertexPositions, std::vector<uint32_t>& indices,
                      std::vector<Vertex>& vertices) {
    const ParsedObj file = parse_obj(modelPath, nullptr);
    append_mesh(ingest(file.attrib, file.shapes, MergeScope::WholeFile).front(), vertexPositions, indices, vertices);
}

// core::file::loadOBJandMTL: one Object + one mesh per shape of the file, materials from the MTL.
inline void loadOBJandMTL(const std::string& objPath, const std::string& mtlPath, std::vector<std::shared_ptr<scene::Object>>& objects,
                          std::vector<vm::vec3>& vertexPositions, std::vector<uint32_t>& indices, std::vector<Vertex>& vertices,
                          std::vector<MeshData>& meshDatas) {
    const ParsedObj file = parse_obj(objPath, mtlPath.empty() ? nullptr : mtlPath.c_str());
    for (const IngestedMesh& m : ingest(file.attrib, file.shapes, MergeScope::PerShape)) {
        const MeshData where = append_mesh(m, vertexPositions, indices, vertices);
        const uint32_t meshIndex = (uint32_t)meshDatas.size();
        meshDatas.push_back(where);          // isOpaque stays false: OBJ/MTL meshes are non-opaque in the reference (quirk Q11)
        objects.emplace_back(new scene::Object(objPath));
        scene::Object& shapeObject = *objects.back();
        const bool hasMaterial = m.firstFaceMaterial >= 0 && m.firstFaceMaterial < (int)file.materials.size();
        apply_material(shapeObject, hasMaterial ? &file.materials[(size_t)m.firstFaceMaterial] : nullptr, mtlPath);
        shapeObject.setBLASIndex(meshIndex);
        shapeObject.setNumTriangles(where.indexCount / 3);
        shapeObject.setVertexOffset(where.vertexIndexOffset);
        shapeObject.setIndexOffset(where.indexIndexOffset);
    }
}

// file.cppm:272-311: stb_image load with vertical flip, RGBA8 (STBI_rgb_alpha) or R8 (STBI_grey).  The decoded
// texels are returned to the caller (who hands them to rtr_scene_create) instead of being uploaded to a vk::Image.
inline rtr::img::Image createTextureImage(const std::string& texturePath, bool isGrayscale) {
    const int channels = isGrayscale ? 1 : 4;
    try {
        return rtr::img::load_image(texturePath, channels, /*flip_vertically=*/true);
    } catch (const std::exception& why) {
        // the caller sees the reference's exception text; the decoder's own diagnosis goes to stderr
        std::cerr << "texture " << texturePath << " could not be decoded (" << why.what() << ")" << std::endl;
        throw std::runtime_error("Image load failed");
    }
}

}  // namespace core::file

namespace app::setup {

namespace vm = rtr::vm;

struct GeometryReturnInfo {
    std::vector<scene::geometry::Vertex> vertices;
    std::vector<vm::vec3> vertexPositions;
    std::vector<uint32_t> indices;
    std::vector<RtrMesh> meshes;         // one per BLAS
    std::vector<RtrInstance> instances;  // TLAS instances: lights, then objects
};

class GeometryBuilder {
public:
    // geometry_builder.cppm:50-212 + tlas.cppm:52-82.  `objects` grows by one Object per OBJ shape of every
    // objMtlPair (as in the reference); offsets / numTriangles / instance indices are written back.
    static GeometryReturnInfo createAccelerationStructures(std::vector<std::shared_ptr<scene::Object>>& objects,
                                                           const std::vector<std::pair<std::string, std::string>>& objMtlPairs,
                                                           std::vector<std::shared_ptr<scene::AreaLight>>& areaLights) {
        GeometryReturnInfo out;
        // 1. the mesh table: every distinct geometry source once.  Lights first, then the explicit objects (a source is named by
        //    its path; "square" is the unit quad an AreaLight carries as points, two triangles 0-1-2, 0-2-3), then the OBJ+MTL
        //    pairs, which add one mesh AND one Object per shape.  A mesh is opaque unless the first instance that brought it in
        //    has an opacity map; OBJ+MTL meshes are never opaque (reference behaviour, see loadOBJandMTL).
        std::vector<core::file::MeshData> table;
        std::unordered_map<std::string, uint32_t> meshOfSource;
        const auto mesh_for = [&](const std::string& source, bool opaque, const std::vector<vm::vec3>& quadPoints) -> uint32_t {
            const auto known = meshOfSource.find(source);
            if (known != meshOfSource.end()) return known->second;
            core::file::MeshData where{};
            if (source == "square") {
                core::file::IngestedMesh quad;
                for (const vm::vec3& p : quadPoints) { scene::geometry::Vertex v{}; v.position = p; quad.vertices.push_back(v); }
                quad.indices = {0u, 1u, 2u, 0u, 2u, 3u};
                where = core::file::append_mesh(quad, out.vertexPositions, out.indices, out.vertices);
            } else {
                where.vertexIndexOffset = (uint32_t)out.vertices.size();
                where.indexIndexOffset = (uint32_t)out.indices.size();
                core::file::loadModel(source, out.vertexPositions, out.indices, out.vertices);
                where.vertexCount = (uint32_t)out.vertices.size() - where.vertexIndexOffset;
                where.indexCount = (uint32_t)out.indices.size() - where.indexIndexOffset;
            }
            where.isOpaque = opaque;
            table.push_back(where);
            meshOfSource.emplace(source, (uint32_t)table.size() - 1);
            return (uint32_t)table.size() - 1;
        };
        for (auto& light : areaLights) light->setBLASIndex(mesh_for(light->getOBJPath(), !light->usesOpacityMap(), light->getPoints()));
        for (auto& object : objects) object->setBLASIndex(mesh_for(object->getOBJPath(), !object->usesOpacityMap(), object->getPoints()));
        for (const auto& pair : objMtlPairs)
            core::file::loadOBJandMTL(pair.first, pair.second, objects, out.vertexPositions, out.indices, out.vertices, table);
        for (const core::file::MeshData& md : table) {
            RtrMesh m{};
            m.vertexOffset = md.vertexIndexOffset; m.indexOffset = md.indexIndexOffset;
            m.vertexCount = md.vertexCount; m.indexCount = md.indexCount; m.isOpaque = md.isOpaque ? 1u : 0u;
            out.meshes.push_back(m);
        }
        // 2. the instances, in the order the hit shaders rely on (behaviour of tlas.cppm:50-82): all lights, then all objects;
        //    customIndex = position in that list.  Each scene item learns its instance index and its mesh's offsets / size.
        const auto instantiate = [&](auto& item) {
            const uint32_t meshIndex = item->getBLASIndex();
            const core::file::MeshData& mesh = table.at(meshIndex);
            RtrInstance inst{};
            inst.meshIndex = meshIndex;
            inst.customIndex = (uint32_t)out.instances.size();
            const vm::Transform34 xf = item->getTransform();
            std::memcpy(inst.transform, xf.matrix, sizeof inst.transform);
            item->setInstanceIndex(inst.customIndex);
            item->setNumTriangles(mesh.indexCount / 3);
            item->setVertexOffset(mesh.vertexIndexOffset);
            item->setIndexOffset(mesh.indexIndexOffset);
            out.instances.push_back(inst);
        };
        for (auto& light : areaLights) instantiate(light);
        for (auto& object : objects) instantiate(object);
        return out;
    }
};

struct SceneReturnInfo {
    std::vector<scene::Object::GPUObjectInfo> GPUObjects;
    std::vector<scene::AreaLight::GPUAreaLightInfo> GPUAreaLights;
    GeometryReturnInfo geoReturnInfo;
    std::vector<rtr::img::Image> textures;        // [0], [1] stay empty: the LTC tables live there in the reference (create_scene.cppm:67-69)
    std::vector<rtr_texture> textureTable;        // C-ABI view of `textures`
    rtr::img::Image hdriImage;                    // application.cppm:250 (optional)
    rtr_texture hdriEntry{};
    void setHDRI(const std::string& path) {
        hdriImage = core::file::createTextureImage(path, false);
        hdriEntry = rtr_texture{hdriImage.pixels.data(), (uint32_t)hdriImage.width, (uint32_t)hdriImage.height, 4u, 0u};
    }
    void rebuildTextureTable() {
        textureTable.clear();
        for (const auto& im : textures)
            textureTable.push_back(rtr_texture{im.pixels.empty() ? nullptr : im.pixels.data(), (uint32_t)im.width, (uint32_t)im.height, (uint32_t)im.channels, 0u});
    }
    // fills the C-ABI descriptor; pointers stay valid while this object lives
    rtr_scene_desc desc(const float* ltc1 = nullptr, const float* ltc2 = nullptr, vm::vec3 sky = vm::vec3(0.f)) const {
        rtr_scene_desc d{};
        static_assert(sizeof(scene::geometry::Vertex) == sizeof(RtrVertex), "layout");
        d.vertices = reinterpret_cast<const RtrVertex*>(geoReturnInfo.vertices.data());
        d.numVertices = (uint32_t)geoReturnInfo.vertices.size();
        d.indices = geoReturnInfo.indices.data(); d.numIndices = (uint32_t)geoReturnInfo.indices.size();
        d.meshes = geoReturnInfo.meshes.data(); d.numMeshes = (uint32_t)geoReturnInfo.meshes.size();
        d.instances = geoReturnInfo.instances.data(); d.numInstances = (uint32_t)geoReturnInfo.instances.size();
        d.objects = GPUObjects.data(); d.numObjects = (uint32_t)GPUObjects.size();
        d.lights = GPUAreaLights.data(); d.numLights = (uint32_t)GPUAreaLights.size();
        d.ltc1 = ltc1; d.ltc2 = ltc2;
        d.skyColor[0] = sky.x; d.skyColor[1] = sky.y; d.skyColor[2] = sky.z;
        d.textures = textureTable.empty() ? nullptr : textureTable.data();
        d.numTextures = (uint32_t)textureTable.size();
        d.hdri = hdriImage.pixels.empty() ? nullptr : &hdriEntry;
        d.buildFlags = RTR_BUILD_HOST_SAH;
        return d;
    }
};

class CreateScene {
public:
    // create_scene.cppm:48-160: geometry, then material textures (path-keyed de-dup, specular -> metallic -> colour ->
    // opacity per object, specular/metallic as R8, colour/opacity as RGBA8, indices starting at 2), then the info arrays.
    static SceneReturnInfo createSceneFromObjectsAndLights(std::vector<std::shared_ptr<scene::Object>>& objects,
                                                           const std::vector<std::pair<std::string, std::string>>& objMtlPairs,
                                                           std::vector<std::shared_ptr<scene::AreaLight>>& areaLights) {
        SceneReturnInfo r;
        r.geoReturnInfo = GeometryBuilder::createAccelerationStructures(objects, objMtlPairs, areaLights);
        r.textures.resize(2);                                                           // LTC1, LTC2 slots (:67-69)
        // a file is decoded once however many objects name it; slots are handed out in the order objects first ask for them,
        // channel by channel: specular, metallic (single-channel), colour, opacity (RGBA)
        std::unordered_map<std::string, uint32_t> slotOfFile;
        const auto slot_for = [&](const std::string& file, bool singleChannel) -> uint32_t {
            const auto known = slotOfFile.find(file);
            if (known != slotOfFile.end()) return known->second;
            const uint32_t slot = (uint32_t)r.textures.size();
            r.textures.push_back(core::file::createTextureImage(file, singleChannel));
            slotOfFile.emplace(file, slot);
            return slot;
        };
        for (const std::shared_ptr<scene::Object>& o : objects) {
            if (o->usesSpecularMap()) o->setSpecularMapIndex(slot_for(o->getSpecularPath(), true));
            if (o->usesMetallicMap()) o->setMetallicMapIndex(slot_for(o->getMetallicPath(), true));
            if (o->usesColorMap()) o->setColorMapIndex(slot_for(o->getColorPath(), false));
            if (o->usesOpacityMap()) o->setOpacityMapIndex(slot_for(o->getOpacityPath(), false));
        }
        r.rebuildTextureTable();
        for (auto& light : areaLights) r.GPUAreaLights.push_back(light->getGPUInfo());   // after TLAS: offsets are set there
        for (auto& object : objects) r.GPUObjects.push_back(object->getGPUInfo());
        return r;
    }
};

}  // namespace app::setup
