// renderer.hpp — C++ RAII shim over the C ABI (include/rtr.h).  Converts every non-zero status into
// std::runtime_error so callers see the reference's error convention (exceptions thrown at the
// failure site, caught once in main: reference src/main.cpp:12-15).  Move-only handles, like the
// reference's vulkan::memory::Buffer (src/vulkan/memory/buffer.cppm:27-30).
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/rtr.h"

namespace rtr {

inline void check(int status, const char* what) {
    if (status != RTR_OK)
        throw std::runtime_error(std::string(what) + ": " + rtr_status_string(status) + ": " + rtr_last_error());
}

class Context {
public:
    explicit Context(int deviceOrdinal = 0) { check(rtr_ctx_create(deviceOrdinal, &h_), "rtr_ctx_create"); }
    ~Context() { if (h_) rtr_ctx_destroy(h_); }
    Context(const Context&) = delete; Context& operator=(const Context&) = delete;
    Context(Context&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    rtr_ctx* get() const { return h_; }
    std::string deviceName() const { char b[256]; check(rtr_ctx_device_name(h_, b, sizeof b), "rtr_ctx_device_name"); return b; }
private:
    rtr_ctx* h_ = nullptr;
};

class Scene {
public:
    Scene(const Context& ctx, const rtr_scene_desc& desc) { check(rtr_scene_create(ctx.get(), &desc, &h_), "rtr_scene_create"); }
    ~Scene() { if (h_) rtr_scene_destroy(h_); }
    Scene(const Scene&) = delete; Scene& operator=(const Scene&) = delete;
    Scene(Scene&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    rtr_scene* get() const { return h_; }
    rtr_scene_stats stats() const { rtr_scene_stats s; check(rtr_scene_get_stats(h_, &s), "rtr_scene_get_stats"); return s; }
private:
    rtr_scene* h_ = nullptr;
};

class Frame {
public:
    Frame(const Context& ctx, uint32_t width, uint32_t rows, uint32_t images) : width_(width), rows_(rows) {
        check(rtr_frame_create(ctx.get(), width, rows, images, &h_), "rtr_frame_create");
    }
    ~Frame() { if (h_) rtr_frame_destroy(h_); }
    Frame(const Frame&) = delete; Frame& operator=(const Frame&) = delete;
    Frame(Frame&& o) noexcept : h_(std::exchange(o.h_, nullptr)), width_(o.width_), rows_(o.rows_) {}
    rtr_frame* get() const { return h_; }
    std::vector<uint32_t> download(int which) const {
        std::vector<uint32_t> px((size_t)width_ * rows_);
        check(rtr_frame_download(h_, which, px.data(), px.size() * 4), "rtr_frame_download");
        return px;
    }
    rtr_frame_stats stats() const { rtr_frame_stats s; check(rtr_frame_get_stats(h_, &s), "rtr_frame_get_stats"); return s; }
    // the passes after the ray-gen dispatch in the reference's frame loop (application.cppm:391-445): a-trous rounds, then combine
    void denoise_combine(int iterations) { check(rtr_denoise_combine(h_, iterations), "rtr_denoise_combine"); }
    uint32_t width() const { return width_; }
    uint32_t rows() const { return rows_; }
private:
    rtr_frame* h_ = nullptr;
    uint32_t width_ = 0, rows_ = 0;
};

inline void render(const Scene& scene, const RtrCameraData& cam, const RtrSceneInfo& info, const rtr_render_params& p, Frame& frame) {
    check(rtr_render(scene.get(), &cam, &info, &p, frame.get()), "rtr_render");
}

}  // namespace rtr
