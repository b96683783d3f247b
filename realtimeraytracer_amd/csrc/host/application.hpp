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

#include "input.hpp"
#include "ltc_tables.hpp"
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
        // The frame the reference PRESENTS (application.cppm:391-457): ray-gen with all five images at NUM_PRIMARY_RAYS = 4, four
        // a-trous rounds on the sampled pair, combine -> FINAL.  Needs the LTC tables (ltcPath; empty -> rtr::ltc::default_path).
        bool present = false;
        std::string ltcPath, dataDir = ".";
        int denoiseIterations = 4;                                          // NUM_DENOISING_ITERATIONS
        // Scripted input, one entry per frame, applied before that frame is rendered — the headless stand-in for Window::processInput
        // and the mouse callback (window.cppm:68-133): keys is any of "WASD" (step = forward / right * camSpeed) and 'T' (toggles the
        // spin, 0.1 degrees of yaw per frame while on); mouseDx / mouseDy are cursor travel (scaled by mouseSensitivity, then by the
        // camera's own 0.1 degrees per unit).
        struct Input { std::string keys; float mouseDx = 0.0f, mouseDy = 0.0f; };
        std::vector<Input> inputs;
        float camSpeed = 10.5f, mouseSensitivity = 0.5f;                    // application.cppm:497-498
    };

    void processInput(const Config::Input& in, float camSpeed, float mouseSensitivity) {
        applyInput(*camera_, in.keys, in.mouseDx, in.mouseDy, camSpeed, mouseSensitivity, spinning_, tWasDown_);
    }
    scene::Camera* camera() { return camera_.get(); }

    Application(std::string_view title, uint32_t width, uint32_t height, bool /*enableValidation*/ = true)
        : title_(title), width_(width), height_(height) {}

    // Returns the last frame's RGBA8 (bytes B,G,R,255) and fills `stats`.
    std::vector<uint32_t> run(Config cfg, rtr_frame_stats* stats = nullptr) {
        camera_ = std::make_unique<scene::Camera>(cfg.fovY, cfg.camPosition, cfg.camLookAt, cfg.camUp, (int)width_, (int)height_);
        auto sceneInfo = app::setup::CreateScene::createSceneFromObjectsAndLights(cfg.objects, cfg.objMtlPairs, cfg.lights);
        rtr::Context ctx(cfg.device);
        rtr::ltc::Tables ltc;
        if (cfg.present) ltc = rtr::ltc::load(cfg.ltcPath.empty() ? rtr::ltc::default_path(cfg.dataDir) : cfg.ltcPath);
        rtr_scene_desc desc = sceneInfo.desc(cfg.present ? ltc.ltc1.data() : nullptr, cfg.present ? ltc.ltc2.data() : nullptr, cfg.sky);
        rtr::Scene scene(ctx, desc);
        const uint32_t renderImages = cfg.present ? RTR_IMAGES_RAYGEN5 : (RTR_IMAGES_FRAMEBUFFER | (cfg.accumulate ? RTR_IMG_BIT(RTR_IMAGE_HDR) : 0u));
        const uint32_t images = cfg.present ? (RTR_IMAGES_RAYGEN5 | RTR_IMAGES_DENOISE) : renderImages;
        rtr::Frame frame(ctx, width_, height_, images);
        spinning_ = false; tWasDown_ = false;
        for (uint32_t frameNo = 0; frameNo < cfg.frames; ++frameNo) {
            if (frameNo < cfg.inputs.size()) processInput(cfg.inputs[frameNo], cfg.camSpeed, cfg.mouseSensitivity);
            camera_->updateGPUData();
            scene::SceneInfo info(frameNo, (uint32_t)cfg.lights.size(), camera_->getPosition());
            rtr_render_params p{};
            p.width = width_; p.height = height_; p.spp = cfg.spp; p.numShadowRays = cfg.numShadowRays;
            p.images = renderImages; p.pipeline = (uint32_t)cfg.pipeline;
            p.accumulate = (cfg.accumulate && !cfg.present) ? 1u : 0u; p.accumulatedFrames = p.accumulate ? frameNo : 0u;
            RtrCameraData cam = camera_->getGPUData();
            rtr::render(scene, cam, info, p, frame);
            if (cfg.present) frame.denoise_combine(cfg.denoiseIterations);
            if (cfg.spinPerFrame != 0.0f) camera_->rotateY(cfg.spinPerFrame);
        }
        std::vector<uint32_t> bgra = frame.download(cfg.present ? RTR_IMAGE_FINAL : RTR_IMAGE_SHADOWED);
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
    bool spinning_ = false, tWasDown_ = false;
};

}  // namespace app
