// application.hpp — headless app::Application: the reference's Application(title, width, height, validation) +
// run() (reference src/app/application.cppm:50-99, 352-480) with the window / swap-chain / present parts removed
// (no display on an MI355X box).  Scene, camera and frame protocol are configured instead of hard-coded
// (the reference has no CLI or config file — SURVEY §5 "Config / flags").
#pragma once
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "renderer.hpp"
#include "scene_builder.hpp"

namespace app {

class Application {
public:
    struct Config {
        std::vector<std::pair<std::string, std::string>> objMtlPairs;       // application.cppm:226-228
        std::vector<std::shared_ptr<scene::AreaLight>> lights;              // application.cppm:184-196
        std::vector<std::shared_ptr<scene::Object>> objects;
        float fovY = 60.0f;                                                 // application.cppm:74-81
        rtr::vm::vec3 camPosition{0.0f, 0.0f, 5.0f}, camLookAt{0.0f, 0.0f, 0.0f}, camUp{0.0f, 1.0f, 0.0f};
        rtr::vm::vec3 sky{0.5f, 0.7f, 1.0f};
        uint32_t spp = 4, numShadowRays = 3;                                // raygen.rgen:8-9
        uint32_t frames = 1;
        bool accumulate = false;                                            // sum frames in float HDR, tonemap once
        int device = 0;
        int pipeline = 0;
        float spinPerFrame = 0.0f;                                          // window.cppm 'T' key: camera.rotateY(0.1) per frame
        std::string outPPM;                                                 // written after the last frame (RGB order)
    };

    Application(std::string_view title, uint32_t width, uint32_t height, bool /*enableValidation*/ = true)
        : title_(title), width_(width), height_(height) {}

    // Returns the last frame's RGBA8 (bytes B,G,R,255) and fills `stats`.
    std::vector<uint32_t> run(Config cfg, rtr_frame_stats* stats = nullptr) {
        camera_ = std::make_unique<scene::Camera>(cfg.fovY, cfg.camPosition, cfg.camLookAt, cfg.camUp, (int)width_, (int)height_);
        auto sceneInfo = app::setup::CreateScene::createSceneFromObjectsAndLights(cfg.objects, cfg.objMtlPairs, cfg.lights);
        rtr::Context ctx(cfg.device);
        rtr_scene_desc desc = sceneInfo.desc(nullptr, nullptr, cfg.sky);
        rtr::Scene scene(ctx, desc);
        const uint32_t images = RTR_IMAGES_FRAMEBUFFER | (cfg.accumulate ? RTR_IMG_BIT(RTR_IMAGE_HDR) : 0u);
        rtr::Frame frame(ctx, width_, height_, images);
        for (uint32_t frameNo = 0; frameNo < cfg.frames; ++frameNo) {
            camera_->updateGPUData();
            scene::SceneInfo info(frameNo, (uint32_t)cfg.lights.size(), camera_->getPosition());
            rtr_render_params p{};
            p.width = width_; p.height = height_; p.spp = cfg.spp; p.numShadowRays = cfg.numShadowRays;
            p.images = images; p.pipeline = (uint32_t)cfg.pipeline;
            p.accumulate = cfg.accumulate ? 1u : 0u; p.accumulatedFrames = cfg.accumulate ? frameNo : 0u;
            RtrCameraData cam = camera_->getGPUData();
            rtr::render(scene, cam, info, p, frame);
            if (cfg.spinPerFrame != 0.0f) camera_->rotateY(cfg.spinPerFrame);
        }
        std::vector<uint32_t> bgra = frame.download(RTR_IMAGE_SHADOWED);
        if (stats) *stats = frame.stats();
        if (!cfg.outPPM.empty()) writePPM(cfg.outPPM, bgra);
        return bgra;
    }

    void writePPM(const std::string& path, const std::vector<uint32_t>& bgra) const {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Failed to open output file: " + path);
        f << "P6 " << width_ << " " << height_ << " 255\n";
        std::vector<unsigned char> rgb((size_t)width_ * height_ * 3);
        for (size_t i = 0; i < (size_t)width_ * height_; ++i) {
            rgb[3 * i + 0] = (unsigned char)((bgra[i] >> 16) & 0xff);   // R is byte 2 (quirk Q14: stored B,G,R,A)
            rgb[3 * i + 1] = (unsigned char)((bgra[i] >> 8) & 0xff);
            rgb[3 * i + 2] = (unsigned char)(bgra[i] & 0xff);
        }
        f.write(reinterpret_cast<const char*>(rgb.data()), (std::streamsize)rgb.size());
    }

private:
    std::string title_;
    uint32_t width_, height_;
    std::unique_ptr<scene::Camera> camera_;
};

}  // namespace app
