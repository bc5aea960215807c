// Main.cpp — the RayTracer application on the GPU path.
// Same flow as reference source/Main.cpp:153-235 (parse, build the Cornell-box
// scene, construct the Renderer with or without a photon map, fill the background,
// render, save the PPM, print the total time); the scene script itself lives in
// ScenePresets.cpp so that tests and bench.py build the identical scene.
#include <chrono>
#include <exception>
#include <iostream>

#include "CommandLine.h"
#include "Image.cpp"
#include "Renderer.cpp"
#include "ScenePresets.h"

int main(int argc, char** argv) {
  CommandLine args;
  try {
    args.parse(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    args.printUsage(argv[0]);
    return 1;
  }
  const auto begin = std::chrono::steady_clock::now();
  GpuSettings::get().device = static_cast<int>(args.gpu());
  GpuSettings::get().seed = static_cast<unsigned>(args.seed());
  GpuSettings::get().accel = args.accel() ? RT_ACCEL_BRUTE : RT_ACCEL_BVH;
  GpuSettings::get().progress = static_cast<unsigned>(args.progress());
  GpuSettings::get().tune = args.tune();
  // -gpus N: devices gpu..gpu+N-1; -devices a,b,c: an explicit list (may repeat a device:
  // rehearsal of the N-rank flow on one GPU)
  if (!args.devices().empty()) {
    size_t pos = 0;
    const std::string& d = args.devices();
    while (pos <= d.size()) {
      const size_t c = d.find(',', pos);
      GpuSettings::get().devices.push_back(std::atoi(d.substr(pos, c == std::string::npos ? c : c - pos).c_str()));
      if (c == std::string::npos) break;
      pos = c + 1;
    }
  } else if (args.gpus() > 1) {
    for (size_t i = 0; i < args.gpus(); ++i) GpuSettings::get().devices.push_back(static_cast<int>(args.gpu() + i));
  }

  try {
    Image image(args.width(), args.height());
    Scene scene = rtpreset::buildCornellScene(args.scene(), args.meshDir(), args.width(), args.height());

    RayTracer rayTracer;
    Renderer renderer;
    if (args.numPhotons() > 0) {
      renderer = Renderer(scene, args.numRays(), args.mode(), rayTracer, args.numPhotons(), args.k());
      renderer.savePhotonMap();
    } else {
      renderer = Renderer(scene, args.numRays(), args.mode(), rayTracer);
    }
    image.fillBackground();
    renderer.render(image);
    image.savePPM(args.outputFilename());

    const rt_stats& st = renderer.lastStats();
    const double rays = static_cast<double>(st.rays_closest + st.rays_shadow);
    std::cout << "GPU: " << rays / 1e6 << " Mrays in " << st.kernel_ms << " ms ("
              << (st.kernel_ms > 0 ? rays / st.kernel_ms / 1e3 : 0.0) << " Mrays/s)" << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  const auto end = std::chrono::steady_clock::now();
  std::cout << "Total time is " << std::chrono::duration_cast<std::chrono::seconds>(end - begin).count() << "[s]"
            << std::endl;
  return 0;
}
