// RayTracer.h — ray casting front end (API of reference source/RayTracer.h:16-118).
//
// rayTrace()/rayTraceBVH() cast ONE ray on the GPU through rt_trace (brute-force
// kernel / BVH kernel respectively; both return the reference's brute-force
// answer).  They exist for API compatibility and tests; the
// renderer never goes through them — it launches the integrator kernel, where
// casting, sampling and shading are fused.  COST: a call flattens the scene and
// creates a device context (BVH build + upload: ~1 ms for the Cornell box, 0.2 s for a
// million triangles) unless the scene is the one of the previous call, then moves one
// ray and one hit record across PCIe — use rt_trace / rt_trace_stream_device with a
// batch of rays for anything that is not a single probe.  The three sampling helpers draw from
// the process-wide engine `gen` with the reference's distributions; inside
// render() the same draws are made on the GPU from per-sample streams.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <random>

#include "Camera.h"
#include "GpuSession.h"
#include "Image.h"
#include "Particle.h"
#include "Scene.h"
#include "Vec3.h"

class RayTracer {
 public:
  RayTracer() {}
  virtual ~RayTracer() {}

  bool rayTrace(const Ray& ray, const Scene& scene, size_t& meshIndex, Vec3i& triangle, float& u, float& v,
                float& d) {
    return cast(ray, scene, RT_ACCEL_BRUTE, meshIndex, triangle, u, v, d);
  }
  bool rayTraceBVH(Ray& ray, const Scene& scene, size_t& nearest_index, Vec3i& nearest_triangle, float& u,
                   float& v, float& d) {
    return cast(ray, scene, RT_ACCEL_BVH, nearest_index, nearest_triangle, u, v, d);
  }

  float stratifiedSample1D(int sampleIdx, int nSamples, float left, float right) {
    std::uniform_real_distribution<> dis(left, right);
    const float inv = (right - left) / nSamples;
    return left + (sampleIdx + 0.5f * dis(gen)) * inv;
  }
  // cell (i % d, i / d) of a d x d grid, d = floor(sqrt(N)), jittered inside the cell
  Vec3f jitterSample(int sampleIdx, int nSamples) {
    std::uniform_real_distribution<> dis(0.0, 1.0);
    const int d = static_cast<int>(std::sqrt(static_cast<float>(nSamples)));
    const float x = static_cast<float>((static_cast<float>(sampleIdx % d) + dis(gen)) / static_cast<float>(d));
    const float y = static_cast<float>((static_cast<float>(sampleIdx / d) + dis(gen)) / static_cast<float>(d));
    return Vec3f(x, y, 0.f);
  }
  Vec3f hsphereUniformSample(Vec3f normal, float maxRayAngle) {
    std::uniform_real_distribution<> dis(0.0, 2 * maxRayAngle / 3.14159265358979323846);
    normal.normalize();
    Vec3f t1, t2;
    normal.getTwoOrthogonals(t1, t2);
    t1.normalize();
    t2.normalize();
    const float theta = static_cast<float>(std::asin(dis(gen)));
    const float phi = static_cast<float>(2 * 3.14159265358979323846 * dis(gen));
    Vec3f ring = t1 * std::cos(phi) + t2 * std::sin(phi);
    ring.normalize();
    return normalize(normal * std::cos(theta) + ring * std::sin(theta));
  }

 private:
  bool cast(const Ray& ray, const Scene& scene, unsigned accel, size_t& meshIndex, Vec3i& triangle, float& u,
            float& v, float& d) {
    // one context per (RayTracer, scene object, geometry size); rebuilt when either changes
    size_t tris = 0;
    for (const Mesh& m : scene.meshes()) tris += m.indexedTriangles().size();
    if (!m_session || m_scene != &scene || m_tris != tris) {
      m_session.reset(new GpuSession(scene, GpuSettings::get().device));
      m_scene = &scene, m_tris = tris;
    }
    rt_ray r;
    for (int c = 0; c < 3; ++c) r.origin[c] = ray.origin()[c], r.direction[c] = ray.direction()[c];
    rt_hit h;
    GpuSession::check(rt_trace(m_session->ctx(), &r, 1, accel, RT_TRACE_CLOSEST, &h), "rt_trace");
    if (!h.hit) return false;
    meshIndex = h.mesh;
    triangle = Vec3i(static_cast<int>(h.vtx[0]), static_cast<int>(h.vtx[1]), static_cast<int>(h.vtx[2]));
    u = h.u, v = h.v, d = h.d;
    return true;
  }
  std::shared_ptr<GpuSession> m_session;
  const Scene* m_scene = nullptr;
  size_t m_tris = 0;
};
