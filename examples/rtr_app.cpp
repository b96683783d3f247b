// rtr_app — headless counterpart of the reference's main() (reference src/main.cpp:7-16): builds an
// app::Application and runs it; any std::exception -> message on stderr + EXIT_FAILURE.
//   rtr_app <scene.obj> <mtl_dir/> <out.ppm> [width height spp frames] [present] [keys=W,WD,,T ...]
// `present`: the frame the reference presents (five ray-gen images, four a-trous rounds, combine) with the shipped LTC tables;
// `keys=`: scripted input, one comma-separated entry per frame (Window::processInput of the reference).
// The camera / light below reproduce realtimeraytracer_amd/scenes.py:cornell_box so the tests can compare outputs.
#include <cstdlib>
#include <iostream>

#include "host/application.hpp"

int main(int argc, char** argv) {
    try {
        if (argc < 4) { std::cerr << "usage: rtr_app scene.obj mtl_dir/ out.ppm [width height spp frames] [present] [keys=W,WD,,T]\n"; return EXIT_FAILURE; }
        const uint32_t w = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 256, h = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 256;
        app::Application app("MI355X ray tracer", w, h, true);
        app::Application::Config cfg;
        cfg.objMtlPairs = {{argv[1], argv[2]}};
        auto light = std::make_shared<scene::AreaLight>(20.0f, rtr::vm::vec3(1.0f, 0.85f, 0.6f), false);
        light->move({278.0f, 547.0f, 279.5f}); light->scale({130.0f, 105.0f, 1.0f}); light->rotate({90.0f, 0.0f, 0.0f});
        cfg.lights.push_back(light);
        cfg.fovY = 40.0f; cfg.camPosition = {278.0f, 273.0f, -800.0f}; cfg.camLookAt = {278.0f, 273.0f, 0.0f};
        cfg.spp = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 1;
        cfg.frames = argc > 7 ? (uint32_t)std::atoi(argv[7]) : 1;
        cfg.outPPM = argv[3];
        for (int i = 8; i < argc; ++i) {
            const std::string opt = argv[i];
            if (opt == "present") {
                cfg.present = true;
                std::string exe = argv[0];                                   // the tables ship next to the executable: <dir>/data/ltc_tables.bin
                const size_t slash = exe.find_last_of('/');
                cfg.dataDir = slash == std::string::npos ? "." : exe.substr(0, slash);
            } else if (opt.rfind("keys=", 0) == 0) {
                std::string rest = opt.substr(5);
                size_t at = 0;
                for (;;) {
                    const size_t comma = rest.find(',', at);
                    app::Application::Config::Input in;
                    in.keys = rest.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
                    cfg.inputs.push_back(in);
                    if (comma == std::string::npos) break;
                    at = comma + 1;
                }
            } else { std::cerr << "unknown option " << opt << "\n"; return EXIT_FAILURE; }
        }
        rtr_frame_stats st{};
        app.run(cfg, &st);
        std::cerr << "rendered " << w << "x" << h << " in " << st.totalMs << " ms (GPU kernels)\n";
        return EXIT_SUCCESS;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
