// Renderer.h — the drop-in boundary: `void Renderer::render(Image&)` (reference
// source/Renderer.h:15-85, source/Renderer.cpp:203-272) dispatching to the GPU.
//
// Same constructors, same render()/savePhotonMap() as the reference.  render():
//   1. flattens the Scene snapshot taken at construction into rt_scene_desc and
//      creates the device context (rt_create: BVH build + upload to HBM);
//   2. with -p > 0: emits photons on the GPU, orders them as the reference's
//      kdtree does, uploads them (rt_emit_photons / rt_host_kd_order / rt_set_photons);
//   3. integrates all spp of all pixels in ONE rt_render call (jitter, camera ray,
//      closest hit, shading, bounces, clamp, accumulate, resolve against the
//      background already in `image`);
//   4. writes update.ppm once, assigns the result to `image`.
// Differences a caller can observe: random numbers come from per-(pixel, sample)
// streams keyed by GpuSettings::seed instead of one global engine (the image is
// statistically, not bitwise, the reference's), and update.ppm is
// written after the last pass only.
#pragma once

#include <string>

#include "GpuSession.h"
#include "PhotonMap.h"
#include "RayTracer.h"
#include "kdtree.h"

#define RAYTRACE 0
#define PATHTRACE 1

class Renderer {
 public:
  Renderer() : m_numRays(0), m_mode(0), m_numPhotons(0), m_k(0) {}
  Renderer(Scene& scene, int numRays, int mode, RayTracer rayTracer)
      : m_numRays(numRays), m_mode(mode), m_numPhotons(0), m_k(0), m_rayTracer(rayTracer), m_scene(scene) {}
  Renderer(Scene& scene, int numRays, int mode, RayTracer rayTracer, int numPhotons, int k)
      : m_numRays(numRays), m_mode(mode), m_numPhotons(numPhotons), m_k(k), m_rayTracer(rayTracer), m_scene(scene) {}
  virtual ~Renderer() {}

  void render(Image& image);
  void savePhotonMap() { m_photonMap.saveToPCD("pointcloud.pcd"); }

  // device-side counters of the last render() (rays cast, kernel time, ...)
  const rt_stats& lastStats() const { return m_stats; }

 private:
  int m_numRays, m_mode, m_numPhotons, m_k;
  PhotonMap m_photonMap, m_importonMap;
  RayTracer m_rayTracer;
  Scene m_scene;
  float m_factor = 100.f;
  rt_stats m_stats = {};
};
