// scene.hpp — the host scene model above the C ABI: scene::Camera, scene::Object, scene::AreaLight,
// scene::SceneInfo, scene::geometry::Vertex with the reference's method names, argument meaning and
// numeric behaviour (quirks included), minus Vulkan types.
//
//   scene::Camera            <- reference src/scene/camera.cppm:19-154  (no Device&, no vk::Buffer:
//                               the 64-B GPUCameraData is handed to rtr_render per frame)
//   scene::Object            <- reference src/scene/object.cppm:18-195
//   scene::AreaLight         <- reference src/scene/area_light.cppm:18-134
//   scene::SceneInfo         <- reference src/scene/scene_info.cppm:10-20
//   scene::geometry::Vertex  <- reference src/scene/geometry/vertex.cppm:11-52
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/rtr_types.h"
#include "vecmath.hpp"

namespace scene {

namespace vm = rtr::vm;

namespace geometry {
struct Vertex {
    vm::vec3 position; float pad0 = 0.0f;
    vm::vec3 normal;   float pad1 = 0.0f;
    vm::vec2 uv;       vm::vec2 pad2;
    bool operator==(const Vertex& o) const {
        return position == o.position && normal == o.normal && uv.x == o.uv.x && uv.y == o.uv.y;
    }
};
static_assert(sizeof(Vertex) == sizeof(RtrVertex), "Vertex layout");
}  // namespace geometry

class Camera {
public:
    using GPUCameraData = RtrCameraData;

    Camera(float fovY, vm::vec3 position, vm::vec3 lookAt, vm::vec3 upVector, int pixelWidth, int pixelHeight)
        : GPUDataNeedsUpdate_(true), position_(position), lookAtPoint_(lookAt), upVector_(upVector), fovY_(fovY),
          pixelWidth_(pixelWidth), pixelHeight_(pixelHeight) {
        // initial yaw / pitch from the lookAt direction (camera.cppm:83-86)
        vm::vec3 dir = vm::normalize(lookAtPoint_ - position_);
        pitch_ = vm::degrees(std::asin(dir.y));
        yaw_ = vm::degrees(std::atan2(dir.z, dir.x));
        updateGPUData();
    }

    void processMouseMovement(float xoffset, float yoffset) {   // camera.cppm:136-148
        constexpr float sensitivity = 0.1f;
        yaw_ += xoffset * sensitivity;
        pitch_ += yoffset * sensitivity;
        if (pitch_ > 89.0f) pitch_ = 89.0f;
        if (pitch_ < -89.0f) pitch_ = -89.0f;
        GPUDataNeedsUpdate_ = true;
    }

    void updateGPUData() {                                       // camera.cppm:98-134
        if (!GPUDataNeedsUpdate_) return;
        float aspect = float(pixelWidth_) / float(pixelHeight_);
        float theta = vm::radians(fovY_);
        float halfHeight = std::tan(theta * 0.5f);
        float halfWidth = aspect * halfHeight;
        float yawRadians = vm::radians(yaw_);
        float pitchRadians = vm::radians(pitch_);
        vm::vec3 direction;
        direction.x = std::cos(pitchRadians) * std::cos(yawRadians);
        direction.y = std::sin(pitchRadians);
        direction.z = std::cos(pitchRadians) * std::sin(yawRadians);
        direction = vm::normalize(direction);
        lookAtPoint_ = position_ + direction;
        vm::vec3 w = vm::normalize(position_ - lookAtPoint_);
        vm::vec3 u = vm::normalize(vm::cross(upVector_, w));
        vm::vec3 v = vm::cross(w, u);
        forward_ = -w;
        right_ = u;
        vm::vec3 hd = (2.0f * halfWidth * u) / float(pixelWidth_);
        vm::vec3 vd = -((2.0f * halfHeight * v) / float(pixelHeight_));
        vm::vec3 tl = position_ - (halfWidth * u) + (halfHeight * v) - w;
        std::memset(&GPUData_, 0, sizeof GPUData_);
        store(GPUData_.position, position_);
        store(GPUData_.horizontalViewportDelta, hd);
        store(GPUData_.verticalViewportDelta, vd);
        store(GPUData_.topLeftViewportCorner, tl);
        GPUDataNeedsUpdate_ = false;
    }

    GPUCameraData getGPUData() { if (GPUDataNeedsUpdate_) updateGPUData(); return GPUData_; }
    vm::vec3 getPosition() { return position_; }
    vm::vec3 getLookAt() { return lookAtPoint_; }
    vm::vec3 getForward() { return forward_; }
    vm::vec3 getRight() { return right_; }
    void setPosition(vm::vec3 p) { position_ = p; GPUDataNeedsUpdate_ = true; }
    void setLookAt(vm::vec3 p) { lookAtPoint_ = p; GPUDataNeedsUpdate_ = true; }
    void rotateY(float angle) { yaw_ += angle; GPUDataNeedsUpdate_ = true; }   // camera.cppm:149-154 (adds to degrees, as there)
    float getYaw() const { return yaw_; }
    float getPitch() const { return pitch_; }

private:
    static void store(float* dst, vm::vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
    bool GPUDataNeedsUpdate_;
    GPUCameraData GPUData_{};
    vm::vec3 position_, lookAtPoint_, upVector_;
    float fovY_;
    int pixelWidth_, pixelHeight_;
    vm::vec3 forward_, right_;
    float yaw_ = -90.0f, pitch_ = 0.0f;
};

// Shared by Object::rotate and AreaLight::rotate (object.cppm:171-195, area_light.cppm:110-134).
// rotation[row][k] indexes a column-major glm::mat3 by [column][row]: the applied matrix is the
// TRANSPOSE of rotZ*rotY*rotX (quirk Q4) — kept, because it is what callers of the API observe.
inline void apply_reference_rotate(vm::Transform34& t, const vm::vec3& degrees) {
    vm::vec3 radians = vm::radians(degrees);
    vm::mat3 rotX = vm::rotation(radians.x, vm::vec3(1, 0, 0));
    vm::mat3 rotY = vm::rotation(radians.y, vm::vec3(0, 1, 0));
    vm::mat3 rotZ = vm::rotation(radians.z, vm::vec3(0, 0, 1));
    vm::mat3 rotation = rotZ * rotY * rotX;
    float result[3][3];
    for (int row = 0; row < 3; ++row)
        for (int col = 0; col < 3; ++col)
            result[row][col] = rotation[row][0] * t.matrix[0][col] + rotation[row][1] * t.matrix[1][col] +
                               rotation[row][2] * t.matrix[2][col];
    for (int row = 0; row < 3; ++row)
        for (int col = 0; col < 3; ++col) t.matrix[row][col] = result[row][col];
}

class Object {
public:
    using GPUObjectInfo = RtrObjectInfo;

    explicit Object(const std::string& objPath) : objPath_(objPath) {}

    void setColor(const std::string& colorPath) { usesColorMap_ = true; colorMap_ = colorPath; }
    void setColor(const vm::vec3 colorVec) { usesColorMap_ = false; colorVec_ = colorVec; }
    void setSpecular(const std::string& p) { usesSpecularMap_ = true; specularMap_ = p; }
    void setSpecular(const float v) { usesSpecularMap_ = false; specularFloat_ = v; }
    void setMetallic(const std::string& p) { usesMetallicMap_ = true; metallicMap_ = p; }
    void setMetallic(const float v) { usesMetallicMap_ = false; metallicFloat_ = v; }
    void setOpacity(const std::string& p) { usesOpacityMap_ = true; opacityMap_ = p; }

    void setColorMapIndex(uint32_t i) { colorMapIndex_ = i; }
    void setSpecularMapIndex(uint32_t i) { specularMapIndex_ = i; }
    void setMetallicMapIndex(uint32_t i) { metallicMapIndex_ = i; }
    void setOpacityMapIndex(uint32_t i) { opacityMapIndex_ = i; }
    void setVertexOffset(uint32_t o) { vertexOffset_ = o; }
    void setIndexOffset(uint32_t o) { indexOffset_ = o; }
    void setNumTriangles(uint32_t n) { numTriangles_ = n; }

    void move(const vm::vec3& m) {                                  // object.cppm:158-162
        transform_.matrix[0][3] += m.x; transform_.matrix[1][3] += m.y; transform_.matrix[2][3] += m.z;
    }
    void scale(float s) {                                           // object.cppm:163-169
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) transform_.matrix[r][c] *= s;
    }
    void rotate(const vm::vec3& degrees) { apply_reference_rotate(transform_, degrees); }

    bool usesColorMap() const { return usesColorMap_; }
    bool usesSpecularMap() const { return usesSpecularMap_; }
    bool usesMetallicMap() const { return usesMetallicMap_; }
    bool usesOpacityMap() const { return usesOpacityMap_; }
    std::string getOBJPath() const { return objPath_; }
    std::string getColorPath() const { return colorMap_; }
    std::string getSpecularPath() const { return specularMap_; }
    std::string getMetallicPath() const { return metallicMap_; }
    std::string getOpacityPath() const { return opacityMap_; }
    void setBLASIndex(uint32_t i) { BLASIndex_ = i; }
    uint32_t getBLASIndex() const { return BLASIndex_; }
    void setInstanceIndex(uint32_t i) { instanceIndex_ = i; }
    uint32_t getInstanceIndex() const { return instanceIndex_; }
    uint32_t getNumTriangles() const { return numTriangles_; }
    vm::Transform34 getTransform() const { return transform_; }
    std::vector<vm::vec3> getPoints() { return {}; }

    GPUObjectInfo getGPUInfo() const {                              // object.cppm:136-155
        GPUObjectInfo g;
        std::memset(&g, 0, sizeof g);
        g.vertexOffset = vertexOffset_; g.indexOffset = indexOffset_;
        g.usesColorMap = usesColorMap_; g.usesSpecularMap = usesSpecularMap_;
        g.usesMetallicMap = usesMetallicMap_; g.usesOpacityMap = usesOpacityMap_;
        g.colorIndex = colorMapIndex_; g.specularIndex = specularMapIndex_;
        g.metallicIndex = metallicMapIndex_; g.opacityIndex = opacityMapIndex_;
        g.color[0] = colorVec_.x; g.color[1] = colorVec_.y; g.color[2] = colorVec_.z;
        g.specular = specularFloat_; g.metallic = metallicFloat_;
        return g;
    }

private:
    const std::string objPath_;
    vm::Transform34 transform_;
    bool usesSpecularMap_ = false, usesMetallicMap_ = false, usesColorMap_ = false, usesOpacityMap_ = false;
    float specularFloat_ = 1.0f, metallicFloat_ = 0.0f;
    vm::vec3 colorVec_ = vm::vec3(0.5f);
    std::string specularMap_, metallicMap_, colorMap_, opacityMap_;
    uint32_t colorMapIndex_ = 0, specularMapIndex_ = 0, metallicMapIndex_ = 0, opacityMapIndex_ = 0;
    uint32_t BLASIndex_ = 0, vertexOffset_ = 0, indexOffset_ = 0, instanceIndex_ = 0, numTriangles_ = 0;
};

class AreaLight {
public:
    using GPUAreaLightInfo = RtrAreaLightInfo;

    AreaLight(float intensity, vm::vec3 color, bool isTwoSided = false, bool isVisible = true,
              const std::string& objPath = "square")
        : intensity_(intensity), color_(color), isTwoSided_(isTwoSided), isVisible_(isVisible), objPath_(objPath) {}

    std::string getOBJPath() const { return objPath_; }
    void move(const vm::vec3& m) {                                  // area_light.cppm:98-102
        transform_.matrix[0][3] += m.x; transform_.matrix[1][3] += m.y; transform_.matrix[2][3] += m.z;
    }
    void scale(vm::vec3 s) {                                        // area_light.cppm:104-108 (diagonal only, quirk Q5)
        transform_.matrix[0][0] *= s.x; transform_.matrix[1][1] *= s.y; transform_.matrix[2][2] *= s.z;
    }
    void rotate(const vm::vec3& degrees) { apply_reference_rotate(transform_, degrees); }

    const std::vector<vm::vec3>& getPoints() const { return points_; }
    void setVertexOffset(uint32_t o) { vertexOffset_ = o; }
    void setIndexOffset(uint32_t o) { indexOffset_ = o; }
    vm::Transform34 getTransform() const { return transform_; }
    void setBLASIndex(uint32_t i) { blasIndex_ = i; }
    uint32_t getBLASIndex() const { return blasIndex_; }
    uint32_t getInstanceIndex() const { return instanceIndex_; }
    void setInstanceIndex(uint32_t i) { instanceIndex_ = i; }
    void setNumTriangles(uint32_t n) { numTriangles_ = n; }
    bool usesOpacityMap() const { return false; }
    bool isVisible() const { return isVisible_; }

    GPUAreaLightInfo getGPUInfo() const {                           // area_light.cppm:85-96 + core/utils.cppm:11-39
        GPUAreaLightInfo g;
        std::memset(&g, 0, sizeof g);
        g.color[0] = color_.x; g.color[1] = color_.y; g.color[2] = color_.z;
        g.intensity = intensity_;
        g.vertexOffset = vertexOffset_; g.indexOffset = indexOffset_; g.numTriangles = numTriangles_;
        g.isTwoSided = isTwoSided_ ? 1u : 0u;
        // PackTransformMatrix: 3x4 row-major -> column-major mat4, last row (0,0,0,1)
        for (int c = 0; c < 4; ++c) {
            for (int r = 0; r < 3; ++r) g.transform[c * 4 + r] = transform_.matrix[r][c];
            g.transform[c * 4 + 3] = (c == 3) ? 1.0f : 0.0f;
        }
        return g;
    }

private:
    float intensity_;
    vm::vec3 color_;
    bool isTwoSided_, isVisible_;
    uint32_t blasIndex_ = 0, instanceIndex_ = 0;
    const std::string objPath_;
    uint32_t vertexOffset_ = 0, indexOffset_ = 0, numTriangles_ = 0;
    vm::Transform34 transform_;
    // initial unit square in XY (area_light.cppm:79-82)
    const std::vector<vm::vec3> points_ = {vm::vec3{-0.5f, -0.5f, 0.0f}, vm::vec3{-0.5f, 0.5f, 0.0f},
                                           vm::vec3{0.5f, 0.5f, 0.0f}, vm::vec3{0.5f, -0.5f, 0.0f}};
};

struct SceneInfo : RtrSceneInfo {                                   // scene_info.cppm:10-20
    SceneInfo(uint32_t frame_, uint32_t num, vm::vec3 cam) {
        frame = frame_; numAreaLights = num; _pad0 = 0; _pad1 = 0;
        camPosition[0] = cam.x; camPosition[1] = cam.y; camPosition[2] = cam.z; pad2_ = 0.0f;
    }
};

}  // namespace scene

namespace std {
template <>
struct hash<scene::geometry::Vertex> {                              // vertex.cppm:29-52
    size_t operator()(const scene::geometry::Vertex& v) const noexcept {
        const float f[8] = {v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.uv.x, v.uv.y};
        size_t seed = std::hash<float>{}(f[0]);
        for (int i = 1; i < 8; ++i) seed ^= std::hash<float>{}(f[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};
}  // namespace std
