// PhotonMap.h — photon list produced by light emission (API of reference
// source/PhotonMap.h:10-156).  Emission runs on the GPU (rt_emit_photons: one lane
// per emitted photon, up to 20 bounces with Russian roulette) with one RNG stream
// per photon; the list order is emission order, as in the reference.
#pragma once

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "GpuSession.h"
#include "Particle.h"
#include "RayTracer.h"
#include "Vec3.h"

class PhotonMap {
 public:
  PhotonMap() {}
  PhotonMap(const Scene& scene, int numOfPhotons, RayTracer /*rayTracer*/) {
    if (numOfPhotons <= 0) return;
    GpuSession session(scene, GpuSettings::get().device);
    emit(session.ctx(), numOfPhotons, scene.lightsources().size());
  }
  // used by Renderer::render, which already owns a session for the scene
  PhotonMap(rt_ctx* ctx, int numOfPhotons, size_t numLights) {
    if (numOfPhotons > 0) emit(ctx, numOfPhotons, numLights);
  }

  int size() { return static_cast<int>(m_list.size()); }
  std::vector<Particle>& list() { return m_list; }
  const std::vector<Particle>& list() const { return m_list; }

  // ASCII PCD v0.7 with fields x y z normal_x normal_y normal_z (PhotonMap.h:59-84)
  void saveToPCD(const std::string& filename) {
    std::ofstream out(filename.c_str());
    if (!out) {
      std::cerr << "Cannot open file " << filename.c_str() << std::endl;
      std::exit(1);
    }
    out << "VERSION .7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\nTYPE F F F F F F\n"
        << "COUNT 1 1 1 1 1 1\nWIDTH " << m_list.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS "
        << m_list.size() << "\nDATA ascii\n";
    for (const Particle& p : m_list)
      out << p.position()[0] << " " << p.position()[1] << " " << p.position()[2] << " " << p.incomeDirection()[0]
          << " " << p.incomeDirection()[1] << " " << p.incomeDirection()[2] << " \n";
    std::cout << "Particle map was saved to: " << filename << std::endl;
  }

 private:
  void emit(rt_ctx* ctx, int numOfPhotons, size_t numLights) {
    std::cout << "Constructing a photon map with " << numOfPhotons << " photons" << std::endl;
    if (numLights)
      std::cout << "Emitting " << static_cast<int>(numOfPhotons * (1.f / numLights)) << " photons per light source"
                << std::endl;
    std::vector<float> pos(3 * static_cast<size_t>(numOfPhotons)), dir(pos.size()), w(numOfPhotons);
    uint32_t n = 0;
    GpuSession::check(rt_emit_photons(ctx, static_cast<uint32_t>(numOfPhotons), GpuSettings::get().seed, pos.data(),
                                      dir.data(), w.data(), &n),
                      "rt_emit_photons");
    m_list.reserve(n);
    for (uint32_t i = 0; i < n; ++i)
      m_list.push_back(Particle(Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                                Vec3f(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), w[i]));
    std::cout << m_list.size() << " photons stored" << std::endl;
  }
  std::vector<Particle> m_list;
};
