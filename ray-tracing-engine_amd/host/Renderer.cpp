// Renderer.cpp — see Renderer.h.  Header-style (include-guarded) because the
// reference application #includes "Renderer.cpp" from Main.cpp (source/Main.cpp:21).
#pragma once
#include "Renderer.h"

#include <iostream>
#include <memory>
#include <vector>

inline void Renderer::render(Image& image) {
  const uint32_t w = static_cast<uint32_t>(image.width()), h = static_cast<uint32_t>(image.height());
  // one device, or (GpuSettings::devices, `-gpus N`) the frame tile-sharded over N of them
  const std::vector<int>& devs = GpuSettings::get().devices;
  const bool multi = devs.size() > 1;
  std::unique_ptr<GpuSession> single;
  std::unique_ptr<GpuGroupSession> many;
  if (multi) many.reset(new GpuGroupSession(m_scene, devs));
  else single.reset(new GpuSession(m_scene, devs.size() == 1 ? devs[0] : GpuSettings::get().device));
  rt_ctx* ctx0 = multi ? many->ctx0() : single->ctx();

  rt_params p = {};
  p.width = w, p.height = h;
  p.spp = static_cast<uint32_t>(m_numRays);
  p.mode = m_mode == PATHTRACE ? RT_MODE_PATH : RT_MODE_RAY;
  p.max_depth = 3;  // calculateColorPath(ray, found, 0, 3) — Renderer.cpp:245,248
  p.seed = GpuSettings::get().seed;
  p.rng_mode = RT_RNG_PIXEL;
  p.accel = GpuSettings::get().accel;
  p.tile = 8;

  // Renderer.cpp:209-213: the photon map and its kd-tree are built inside render();
  // the map is a local there (the member stays empty, so a savePhotonMap() issued
  // BEFORE render() writes an empty cloud, as in the reference).  We keep the member
  // filled afterwards (in tree order) so that a later savePhotonMap() is useful.
  if (m_numPhotons > 0) {
    // the whole map on the device(s): emission, compaction and the kd order (the
    // reference's std::nth_element order, restated tie-exactly: csrc/kd_build.hip); every
    // device of a group builds the same map from the same streams.  The host copy only
    // feeds a later savePhotonMap().
    std::cout << "Constructing a photon map with " << m_numPhotons << " photons" << std::endl;
    if (!m_scene.lightsources().empty())
      std::cout << "Emitting " << static_cast<int>(m_numPhotons * (1.f / m_scene.lightsources().size()))
                << " photons per light source" << std::endl;
    std::cout << "Constructing a kd-tree for the photon map." << std::endl;
    uint32_t stored = 0;
    const uint32_t ranks = multi ? rt_group_size(many->group()) : 1u;
    for (uint32_t r = 0; r < ranks; ++r) {
      rt_ctx* cr = multi ? rt_group_ctx(many->group(), r) : ctx0;
      GpuSession::check(rt_build_photon_map(cr, static_cast<uint32_t>(m_numPhotons), GpuSettings::get().seed, &stored, nullptr),
                        "rt_build_photon_map");
    }
    std::cout << stored << " photons stored" << std::endl;
    if (stored) {
      std::vector<float> pos(3 * static_cast<size_t>(stored)), dir(pos.size()), wt(stored);
      uint32_t n = 0;
      GpuSession::check(rt_get_photons(ctx0, pos.data(), dir.data(), wt.data(), stored, &n), "rt_get_photons");
      PhotonMap map;
      for (uint32_t i = 0; i < n; ++i)
        map.list().push_back(Particle(Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                                      Vec3f(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), wt[i]));
      m_photonMap = map;
      p.use_photons = 1, p.k = static_cast<uint32_t>(m_k), p.photons_requested = static_cast<uint32_t>(m_numPhotons);
    }
  }

  // -tune S (extension): the tree tuned on probe frames of THIS camera and integrator (rt_bvh_tune: same picture,
  // fewer node visits; pays off on small scenes rendered with many samples)
  if (GpuSettings::get().tune > 0. && !multi && !p.use_photons && p.accel == RT_ACCEL_BVH && p.spp > 0) {
    rt_params probe = p;
    const uint32_t longer = w > h ? w : h, scale = longer > 128u ? (longer + 127u) / 128u : 1u;
    probe.width = (w + scale - 1u) / scale, probe.height = (h + scale - 1u) / scale;
    probe.spp = 1;
    rt_tune_report rep = {};
    GpuSession::check(rt_bvh_tune(ctx0, &probe, GpuSettings::get().tune, 0u, &rep), "rt_bvh_tune");
    std::cout << "BVH tuned on " << probe.width << "x" << probe.height << " probe frames: " << rep.probes << " probes, " << rep.accepted
              << " changes kept, measured cost " << rep.cost_before << " -> " << rep.cost_after << " in " << rep.seconds << " s" << std::endl;
  }

  Image result(w, h);
  if (p.spp > 0 && multi) {
    // N devices: the whole frame in one sharded launch per device (rt_group_render);
    // update.ppm is written once, after the last pass
    p.tile = 32;
    rt_stats st = {};
    GpuSession::check(rt_group_render(many->group(), &p, image.data(), result.data(), nullptr, &st), "rt_group_render");
    m_stats = st;
    result.savePPM("update.ppm");
    std::cout << "Raytracing on " << devs.size() << " GPUs" << (rt_group_uses_rccl(many->group()) ? " (RCCL)" : " (peer copies)")
              << "... [" << std::string(50, '#') << "] 100%" << std::endl;
    image = result;
  } else if (p.spp > 0) {
    // The reference re-saves update.ppm after EVERY pass (Renderer.cpp:261-269).  Here a
    // "pass" is a sample range of one launch; GpuSettings::progress = P > 0 renders P
    // samples per launch and saves the running estimate after each (resolved with the
    // number of samples so far, as the reference does), 0 renders the frame in one launch.
    // Sample ranges never change the result (the per-pixel sum order is fixed).
    const uint32_t chunk = GpuSettings::get().progress ? GpuSettings::get().progress : p.spp;
    std::vector<float> accum(static_cast<size_t>(w) * h * 4, 0.f);
    rt_stats total = {};
    for (uint32_t done = 0; done < p.spp; done += chunk) {
      p.spp_begin = done;
      p.spp_count = done + chunk > p.spp ? p.spp - done : chunk;
      rt_stats st = {};
      GpuSession::check(rt_render_passes(ctx0, &p, image.data(), accum.data(), result.data(), &st), "rt_render");
      total.samples += st.samples, total.rays_closest += st.rays_closest, total.rays_shadow += st.rays_shadow;
      total.knn_queries += st.knn_queries, total.kernel_ms += st.kernel_ms;
      result.savePPM("update.ppm");
    }
    m_stats = total;
    std::cout << "Raytracing... [" << std::string(50, '#') << "] 100%" << std::endl;
    image = result;
  }
}
