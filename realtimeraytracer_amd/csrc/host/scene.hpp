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
#include <algorithm>
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
        // the angles the camera starts from are those of the unit vector towards the look-at point (behaviour of camera.cppm:83-86)
        const vm::vec3 towards = vm::normalize(lookAtPoint_ - position_);
        yaw_ = vm::degrees(std::atan2(towards.z, towards.x));
        pitch_ = vm::degrees(std::asin(towards.y));
        updateGPUData();
    }

    // Mouse look (behaviour of camera.cppm:136-148): a tenth of a degree per unit of mouse travel; pitch stops one degree short of
    // straight up / down.
    void processMouseMovement(float xoffset, float yoffset) {
        const float degreesPerUnit = 0.1f, pitchLimit = 89.0f;
        yaw_ += xoffset * degreesPerUnit;
        pitch_ = std::min(pitchLimit, std::max(-pitchLimit, pitch_ + yoffset * degreesPerUnit));
        GPUDataNeedsUpdate_ = true;
    }

    // Behaviour of camera.cppm:98-134: the view direction follows from yaw / pitch (degrees), the look-at point is re-derived from
    // it, and the 64-B record holds the eye, the world-space step per pixel along the image's right (+x) and down (+y) axes, and the
    // world-space position of the image's top-left corner on the plane one unit in front of the eye.  The float operations and
    // their order are the reference's (the known answers of SURVEY Appendix B pin them); the code is organised around a small
    // orthonormal frame instead.
    struct ViewFrame { vm::vec3 back, right, up; };           // back = away from what is looked at

    static vm::vec3 directionFromAngles(float yawDegrees, float pitchDegrees) {
        const float yaw = vm::radians(yawDegrees), pitch = vm::radians(pitchDegrees);
        vm::vec3 d;
        d.x = std::cos(pitch) * std::cos(yaw);
        d.y = std::sin(pitch);
        d.z = std::cos(pitch) * std::sin(yaw);
        return vm::normalize(d);
    }

    ViewFrame frameTowards(vm::vec3 target) const {
        ViewFrame f;
        f.back = vm::normalize(position_ - target);
        f.right = vm::normalize(vm::cross(upVector_, f.back));
        f.up = vm::cross(f.back, f.right);
        return f;
    }

    void updateGPUData() {
        if (!GPUDataNeedsUpdate_) return;
        lookAtPoint_ = position_ + directionFromAngles(yaw_, pitch_);
        const ViewFrame f = frameTowards(lookAtPoint_);
        forward_ = -f.back;
        right_ = f.right;
        // half extents of the image plane at distance 1: vertical from the field of view, horizontal through the pixel aspect
        const float halfV = std::tan(vm::radians(fovY_) * 0.5f);
        const float halfH = (float(pixelWidth_) / float(pixelHeight_)) * halfV;
        std::memset(&GPUData_, 0, sizeof GPUData_);
        store(GPUData_.position, position_);
        store(GPUData_.horizontalViewportDelta, (2.0f * halfH * f.right) / float(pixelWidth_));
        store(GPUData_.verticalViewportDelta, -((2.0f * halfV * f.up) / float(pixelHeight_)));
        store(GPUData_.topLeftViewportCorner, position_ - (halfH * f.right) + (halfV * f.up) - f.back);
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

// Object::rotate / AreaLight::rotate (behaviour of object.cppm:171-195, area_light.cppm:110-134): Euler angles in degrees, composed
// as Z * Y * X.  What reaches the transform is the TRANSPOSE of that product (quirk Q4: the reference indexes a column-major
// matrix as [row][k]) — the inverse rotation — applied from the left to the 3x3 linear part; the translation column is left
// alone.  Kept because it is what callers of the API observe; here the transpose is taken explicitly.
inline void apply_reference_rotate(vm::Transform34& t, const vm::vec3& degrees) {
    const vm::vec3 a = vm::radians(degrees);
    const vm::mat3 zyx = vm::rotation(a.z, vm::vec3(0, 0, 1)) * vm::rotation(a.y, vm::vec3(0, 1, 0)) * vm::rotation(a.x, vm::vec3(1, 0, 0));
    float applied[3][3];                           // applied[i][k] = element (row k, column i) of zyx = transpose(zyx)[i][k]
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) applied[i][k] = zyx[i][k];        // zyx[i] is COLUMN i, so zyx[i][k] is row k of column i
    float linear[3][3];
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c)
            linear[i][c] = applied[i][0] * t.matrix[0][c] + applied[i][1] * t.matrix[1][c] + applied[i][2] * t.matrix[2][c];
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c) t.matrix[i][c] = linear[i][c];
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

    void move(const vm::vec3& m) {                                  // adds to the translation column (object.cppm:158-162)
        for (int axis = 0; axis < 3; ++axis) transform_.matrix[axis][3] += m[axis];
    }
    void scale(float s) {                                           // uniform, linear part only (object.cppm:163-169)
        for (auto& row : transform_.matrix) for (int c = 0; c < 3; ++c) row[c] *= s;
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
    void move(const vm::vec3& m) {                                  // adds to the translation column (area_light.cppm:98-102)
        for (int axis = 0; axis < 3; ++axis) transform_.matrix[axis][3] += m[axis];
    }
    void scale(vm::vec3 s) {                                        // the DIAGONAL only, whatever the rotation so far (quirk Q5, area_light.cppm:104-108)
        for (int axis = 0; axis < 3; ++axis) transform_.matrix[axis][axis] *= s[axis];
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
struct hash<scene::geometry::Vertex> {                              // the de-duplication key of vertex.cppm:29-52 (any hash will do)
    size_t operator()(const scene::geometry::Vertex& v) const noexcept {
        // FNV-1a over the bit patterns of the eight floats (-0.0 folded onto 0.0 so that equal vertices hash alike)
        const float f[8] = {v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.uv.x, v.uv.y};
        uint64_t h = 1469598103934665603ull;
        for (float x : f) {
            uint32_t bits; const float y = x == 0.0f ? 0.0f : x; std::memcpy(&bits, &y, sizeof bits);
            for (int b = 0; b < 4; ++b) { h ^= (bits >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
        }
        return (size_t)h;
    }
};
}  // namespace std
