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
  // filled afterwards so that a later savePhotonMap() is useful.
  if (m_numPhotons > 0) {
    PhotonMap map(ctx0, m_numPhotons, m_scene.lightsources().size());
    std::cout << "Constructing a kd-tree for the photon map." << std::endl;
    kdtree tree(map.list().begin(), map.list().end());
    if (!tree.empty()) {
      std::vector<float> pos(3 * tree.size()), dir(3 * tree.size());
      for (size_t i = 0; i < tree.size(); ++i)
        for (int c = 0; c < 3; ++c)
          pos[3 * i + c] = tree.nodes()[i].position()[c], dir[3 * i + c] = tree.nodes()[i].incomeDirection()[c];
      if (multi)
        GpuSession::check(rt_group_set_photons(many->group(), pos.data(), dir.data(), static_cast<uint32_t>(tree.size())),
                          "rt_group_set_photons");
      else
        GpuSession::check(rt_set_photons(ctx0, pos.data(), dir.data(), static_cast<uint32_t>(tree.size())), "rt_set_photons");
      p.use_photons = 1, p.k = static_cast<uint32_t>(m_k), p.photons_requested = static_cast<uint32_t>(m_numPhotons);
    }
    m_photonMap = map;
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
