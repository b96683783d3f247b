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

inline Vertex make_vertex(const rtr::obj::attrib_t& attrib, const rtr::obj::index_t& index) {
    // the reference indexes attrib blindly (file.cppm:151-183); a face that names a vertex / normal / uv the file does not
    // define is refused here instead of reading out of bounds
    if (index.vertex_index < 0 || 3 * (size_t)index.vertex_index + 2 >= attrib.vertices.size() ||
        (index.normal_index >= 0 && 3 * (size_t)index.normal_index + 2 >= attrib.normals.size()) ||
        (index.texcoord_index >= 0 && 2 * (size_t)index.texcoord_index + 1 >= attrib.texcoords.size()))
        throw std::runtime_error("OBJ: a face references a vertex, normal or texture coordinate that is not defined");
    Vertex vertex{};
    vertex.position = {attrib.vertices[3 * index.vertex_index + 0], attrib.vertices[3 * index.vertex_index + 1],
                       attrib.vertices[3 * index.vertex_index + 2]};
    if (index.normal_index >= 0)
        vertex.normal = {attrib.normals[3 * index.normal_index + 0], attrib.normals[3 * index.normal_index + 1],
                         attrib.normals[3 * index.normal_index + 2]};
    else
        vertex.normal = vm::vec3(0.0f, 0.0f, 0.0f);
    if (index.texcoord_index >= 0) {
        vertex.uv.x = attrib.texcoords[2 * index.texcoord_index + 0];
        vertex.uv.y = attrib.texcoords[2 * index.texcoord_index + 1];
    }
    return vertex;
}

// file.cppm:44-102: the whole file becomes ONE mesh, vertices de-duplicated across all shapes.
inline void loadModel(const std::string& modelPath, std::vector<vm::vec3>& vertexPositions, std::vector<uint32_t>& indices,
                      std::vector<Vertex>& vertices) {
    rtr::obj::attrib_t attrib; std::vector<rtr::obj::shape_t> shapes; std::vector<rtr::obj::material_t> materials;
    std::string warn, err;
    uint32_t newVertexCount = 0;
    if (!rtr::obj::LoadObj(&attrib, &shapes, &materials, &warn, &err, modelPath.c_str())) throw std::runtime_error(warn + err);
    std::unordered_map<Vertex, uint32_t> uniqueVertices{};
    for (const auto& shape : shapes)
        for (const auto& index : shape.mesh.indices) {
            Vertex vertex = make_vertex(attrib, index);
            if (uniqueVertices.count(vertex) == 0) {
                uniqueVertices[vertex] = newVertexCount;
                vertices.push_back(vertex);
                vertexPositions.push_back(vertex.position);
                ++newVertexCount;
            }
            indices.push_back(uniqueVertices[vertex]);
        }
}

inline std::string normalizePath(std::string path) { std::replace(path.begin(), path.end(), '\\', '/'); return path; }

// file.cppm:112-269: one Object + one mesh per OBJ shape; material of the shape's FIRST face (quirk Q12).
inline void loadOBJandMTL(const std::string& objPath, const std::string& mtlPath, std::vector<std::shared_ptr<scene::Object>>& objects,
                          std::vector<vm::vec3>& vertexPositions, std::vector<uint32_t>& indices, std::vector<Vertex>& vertices,
                          std::vector<MeshData>& meshDatas) {
    rtr::obj::attrib_t attrib; std::vector<rtr::obj::shape_t> shapes; std::vector<rtr::obj::material_t> materials;
    std::string warn, err;
    if (!rtr::obj::LoadObj(&attrib, &shapes, &materials, &warn, &err, objPath.c_str(), mtlPath.empty() ? nullptr : mtlPath.c_str()))
        throw std::runtime_error(warn + err);
    for (size_t s = 0; s < shapes.size(); ++s) {
        const auto& shape = shapes[s];
        auto obj = std::make_shared<scene::Object>(objPath);
        std::vector<Vertex> shapeVertices; std::vector<uint32_t> shapeIndices;
        std::unordered_map<Vertex, uint32_t> uniqueVertices{};
        uint32_t nextIndex = 0;
        size_t index_offset = 0;
        for (size_t f = 0; f < shape.mesh.num_face_vertices.size(); f++) {
            int fv = (int)shape.mesh.num_face_vertices[f];
            if (fv != 3) {
                std::cerr << "Non-triangle face detected, skipping face." << std::endl;
                index_offset += fv;
                continue;
            }
            for (int v = 0; v < fv; ++v) {
                Vertex vertex = make_vertex(attrib, shape.mesh.indices[index_offset + v]);
                if (uniqueVertices.count(vertex) == 0) { uniqueVertices[vertex] = nextIndex++; shapeVertices.push_back(vertex); }
                shapeIndices.push_back(uniqueVertices[vertex]);
            }
            index_offset += fv;
        }
        size_t prevVertexCount = vertexPositions.size(), prevIndexCount = indices.size();
        for (const auto& vertex : shapeVertices) { vertexPositions.push_back(vertex.position); vertices.push_back(vertex); }
        for (const auto& index : shapeIndices) indices.push_back(index);
        size_t newVertexCount = vertexPositions.size() - prevVertexCount, newIndexCount = indices.size() - prevIndexCount;

        int matId = shape.mesh.material_ids.empty() ? -1 : shape.mesh.material_ids[0];
        if (matId >= 0 && matId < (int)materials.size()) {
            const auto& mat = materials[matId];
            if (!mat.diffuse_texname.empty()) obj->setColor(normalizePath(mtlPath + mat.diffuse_texname));
            else obj->setColor(vm::vec3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]));
            if (!mat.specular_texname.empty()) obj->setSpecular(normalizePath(mtlPath + mat.specular_texname));
            else obj->setSpecular(mat.specular[0]);
            if (!mat.metallic_texname.empty()) obj->setMetallic(normalizePath(mtlPath + mat.metallic_texname));
            else if (mat.unknown_parameter.find("metallic") != mat.unknown_parameter.end())
                obj->setMetallic(std::stof(mat.unknown_parameter.at("metallic")));
            else obj->setMetallic(0.0f);
            if (!mat.alpha_texname.empty()) obj->setOpacity(normalizePath(mtlPath + mat.alpha_texname));
        } else {
            obj->setColor(vm::vec3(0.5f)); obj->setSpecular(1.0f); obj->setMetallic(0.0f);
        }
        obj->setVertexOffset((uint32_t)prevVertexCount);
        obj->setIndexOffset((uint32_t)prevIndexCount);
        obj->setNumTriangles((uint32_t)(newIndexCount / 3));
        obj->setBLASIndex((uint32_t)meshDatas.size());
        objects.push_back(std::move(obj));
        MeshData md{};
        md.vertexIndexOffset = (uint32_t)prevVertexCount; md.indexIndexOffset = (uint32_t)prevIndexCount;
        md.vertexCount = (uint32_t)newVertexCount; md.indexCount = (uint32_t)newIndexCount;
        meshDatas.push_back(md);
    }
}

// file.cppm:272-311: stb_image load with vertical flip, RGBA8 (STBI_rgb_alpha) or R8 (STBI_grey).  The decoded
// texels are returned to the caller (who hands them to rtr_scene_create) instead of being uploaded to a vk::Image.
inline rtr::img::Image createTextureImage(const std::string& texturePath, bool isGrayscale) {
    try {
        return rtr::img::load_image(texturePath, isGrayscale ? 1 : 4, /*flip_vertically=*/true);
    } catch (const std::exception& e) {
        std::cerr << "Failed to load image: " << texturePath << "\n";
        std::cerr << "Reason: " << e.what() << "\n";
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
        std::vector<core::file::MeshData> meshDatas;
        std::unordered_map<std::string, int> loadedModels;
        auto processGeometryObject = [&](auto& instance) {
            std::string modelPath = instance->getOBJPath();
            auto it = loadedModels.find(modelPath);
            if (it != loadedModels.end()) { instance->setBLASIndex(it->second); return; }
            size_t prevVertexCount = out.vertexPositions.size(), prevIndexCount = out.indices.size();
            if (modelPath == "square") {
                for (const vm::vec3& point : instance->getPoints()) {
                    out.vertexPositions.push_back(point);
                    scene::geometry::Vertex v{}; v.position = point;
                    out.vertices.push_back(v);
                }
                for (uint32_t index : {0u, 1u, 2u, 0u, 2u, 3u}) out.indices.push_back(index);
            } else {
                core::file::loadModel(modelPath, out.vertexPositions, out.indices, out.vertices);
            }
            core::file::MeshData md{};
            md.vertexCount = (uint32_t)(out.vertexPositions.size() - prevVertexCount);
            md.indexCount = (uint32_t)(out.indices.size() - prevIndexCount);
            md.vertexIndexOffset = (uint32_t)prevVertexCount; md.indexIndexOffset = (uint32_t)prevIndexCount;
            md.isOpaque = !instance->usesOpacityMap();
            meshDatas.push_back(md);
            instance->setBLASIndex((uint32_t)meshDatas.size() - 1);
            loadedModels[modelPath] = (int)meshDatas.size() - 1;
        };
        for (auto& light : areaLights) processGeometryObject(light);     // lights always go first
        for (auto& object : objects) processGeometryObject(object);
        for (const auto& [objPath, mtlPath] : objMtlPairs)
            core::file::loadOBJandMTL(objPath, mtlPath, objects, out.vertexPositions, out.indices, out.vertices, meshDatas);

        for (const auto& md : meshDatas) {
            RtrMesh m{};
            m.vertexOffset = md.vertexIndexOffset; m.indexOffset = md.indexIndexOffset;
            m.vertexCount = md.vertexCount; m.indexCount = md.indexCount; m.isOpaque = md.isOpaque ? 1u : 0u;
            out.meshes.push_back(m);
        }
        // TLAS::TLAS (tlas.cppm:50-82)
        uint32_t i = 0;
        auto createBLASInstance = [&](auto& currObject) {
            uint32_t blasIndex = currObject->getBLASIndex();
            const core::file::MeshData& blas = meshDatas.at(blasIndex);
            currObject->setNumTriangles(blas.indexCount / 3);
            RtrInstance inst{};
            vm::Transform34 t = currObject->getTransform();
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) inst.transform[r * 4 + c] = t.matrix[r][c];
            inst.meshIndex = blasIndex;
            currObject->setInstanceIndex(i);
            inst.customIndex = i;
            currObject->setVertexOffset(blas.vertexIndexOffset);
            currObject->setIndexOffset(blas.indexIndexOffset);
            out.instances.push_back(inst);
            ++i;
        };
        for (auto& light : areaLights) createBLASInstance(light);
        for (auto& object : objects) createBLASInstance(object);
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
        std::unordered_map<std::string, int> loadedTextures;                            // :71
        auto useTexture = [&](const std::string& path, bool gray) -> uint32_t {
            auto it = loadedTextures.find(path);
            if (it != loadedTextures.end()) return (uint32_t)it->second;
            r.textures.push_back(core::file::createTextureImage(path, gray));
            loadedTextures[path] = (int)r.textures.size() - 1;
            return (uint32_t)r.textures.size() - 1;
        };
        for (auto& object : objects) {                                                  // :75-136
            if (object->usesSpecularMap()) object->setSpecularMapIndex(useTexture(object->getSpecularPath(), true));
            if (object->usesMetallicMap()) object->setMetallicMapIndex(useTexture(object->getMetallicPath(), true));
            if (object->usesColorMap()) object->setColorMapIndex(useTexture(object->getColorPath(), false));
            if (object->usesOpacityMap()) object->setOpacityMapIndex(useTexture(object->getOpacityPath(), false));
        }
        r.rebuildTextureTable();
        for (auto& light : areaLights) r.GPUAreaLights.push_back(light->getGPUInfo());   // after TLAS: offsets are set there
        for (auto& object : objects) r.GPUObjects.push_back(object->getGPUInfo());
        return r;
    }
};

}  // namespace app::setup
