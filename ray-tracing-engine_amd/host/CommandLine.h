// CommandLine.h — argument parser of the RayTracer application.
// Flag set, defaults and error behaviour of reference source/CommandLine.h:8-102
// (-w/-width -h/-height -o/-output -N/-n/-numRays -m/-mode -p/-numPhotons -k -help;
// 380x270, N=16, mode 0, no photons, k=5, output.ppm; a trailing flag without a
// value is "Missing argument" unless it is -help; unknown flags throw; a mode other
// than 0/1 silently becomes 0), plus build-defined extensions the reference has
// no equivalent for: -scene, -meshdir, -seed, -gpu, -gpus, -devices, -accel,
// -progress, -tune.
#pragma once

#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

using namespace std;  // the reference header leaks this; user code relies on it

class CommandLine {
 public:
  CommandLine()
      : m_width(380), m_height(270), m_numRays(16), m_mode(0), m_numPhotons(0), m_k(5), m_seed(1), m_gpu(0),
        m_accel(0), m_progress(0), m_gpus(1), m_tune(0.), m_outputFilename("output.ppm"), m_scene("cubes"), m_meshDir("../meshes") {}
  virtual ~CommandLine() {}

  size_t width() const { return m_width; }
  size_t height() const { return m_height; }
  size_t numRays() const { return m_numRays; }
  size_t mode() const { return m_mode; }
  size_t numPhotons() const { return m_numPhotons; }
  size_t k() const { return m_k; }
  const std::string& outputFilename() const { return m_outputFilename; }
  // extensions
  size_t seed() const { return m_seed; }
  size_t gpu() const { return m_gpu; }
  size_t accel() const { return m_accel; }
  size_t progress() const { return m_progress; }
  double tune() const { return m_tune; }
  size_t gpus() const { return m_gpus; }
  const std::string& devices() const { return m_devices; }
  const std::string& scene() const { return m_scene; }
  const std::string& meshDir() const { return m_meshDir; }

  void printUsage(const char* command) {
    std::cerr << "USAGE: " << command
              << " [-w/-width <image width>][-h/-height <image height>][-o/-output <outputfilename>]"
                 "[-N/-n/-numRays <number of rays per pixel>][-m/-mode <mode (0 for Ray tracing, 1 for Path "
                 "tracing)>][-p/-numPhotons <number of photons for a photon map. If defined, photon map-based "
                 "rendering is used.>][-k <number of neighbours in photon mapping. Use only with -p/-numPhotons>]"
                 "[-scene <cubes|lowres|hires|stress|file:NAME.off>][-meshdir <dir with .off files>]"
                 "[-seed <per-pixel RNG stream key>][-gpu <HIP device>][-gpus <N: tile-shard the frame over devices gpu..gpu+N-1>]"
                 "[-devices <a,b,c: explicit device list for -gpus>][-accel <0 BVH | 1 brute force>]"
                 "[-progress <samples per update.ppm; 1 = after every pass like the reference, 0 = once>]"
                 "[-tune <seconds: tune the BVH on probe frames of this camera before rendering (rt_bvh_tune); 0 = off>]"
              << std::endl;
  }

  void parse(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
      const std::string flag = argv[i];
      if (i == argc - 1) {
        if (flag == "-help") {
          printUsage(argv[0]);
          std::exit(0);
        }
        throw std::runtime_error("Missing argument");
      }
      const char* value = argv[++i];
      if (flag == "-w" || flag == "-width") m_width = std::atoi(value);
      else if (flag == "-h" || flag == "-height") m_height = std::atoi(value);
      else if (flag == "-o" || flag == "-output") m_outputFilename = value;
      else if (flag == "-N" || flag == "-n" || flag == "-numRays") m_numRays = std::atoi(value);
      else if (flag == "-m" || flag == "-mode") m_mode = std::atoi(value);
      else if (flag == "-p" || flag == "-numPhotons") m_numPhotons = std::atoi(value);
      else if (flag == "-k") m_k = std::atoi(value);
      else if (flag == "-scene") m_scene = value;
      else if (flag == "-meshdir") m_meshDir = value;
      else if (flag == "-seed") m_seed = std::atoi(value);
      else if (flag == "-gpu") m_gpu = std::atoi(value);
      else if (flag == "-gpus") m_gpus = std::atoi(value);
      else if (flag == "-devices") m_devices = value;
      else if (flag == "-accel") m_accel = std::atoi(value);
      else if (flag == "-progress") m_progress = std::atoi(value);
      else if (flag == "-tune") m_tune = std::atof(value);
      else throw std::runtime_error("Unknown argument <" + flag + ">");
    }
    if (m_mode != 0 && m_mode != 1) m_mode = 0;
    cout << "#########################" << endl;
    cout << "Mode: " << (m_mode == 1 ? "Path tracing" : "Ray tracing") << endl;
    cout << "Photon map ";
    if (m_numPhotons == 0) cout << "OFF" << endl;
    else cout << "ON with " << m_numPhotons << " photons. Number of searched neighbours equals " << m_k << endl;
    cout << "width: " << m_width << ", height: " << m_height << endl;
    cout << "Output image filename: " << m_outputFilename << endl;
    cout << "#########################" << endl << endl;
  }

 private:
  size_t m_width, m_height, m_numRays, m_mode, m_numPhotons, m_k, m_seed, m_gpu, m_accel, m_progress, m_gpus;
  double m_tune;
  std::string m_outputFilename, m_scene, m_meshDir, m_devices;
};
