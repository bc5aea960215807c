// GpuSession.h — RAII owner of an rt_ctx for the host classes.  Every host class
// that has to EVALUATE something (Renderer, RayTracer, PhotonMap, kdtree) goes
// through one of these; construction throws std::runtime_error with the C ABI's
// message when the HIP library cannot reach a gfx950 device — there is no host
// implementation to fall back to.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "SceneFlatten.h"
#include "rt_amd.h"

class GpuSession {
 public:
  GpuSession() : m_ctx(nullptr) {}
  explicit GpuSession(const Scene& scene, int device = 0) : m_flat(scene), m_ctx(nullptr) { open(device); }
  ~GpuSession() { close(); }
  GpuSession(const GpuSession&) = delete;
  GpuSession& operator=(const GpuSession&) = delete;

  void reset(const Scene& scene, int device = 0) {
    close();
    m_flat.assign(scene);
    open(device);
  }
  void close() {
    if (m_ctx) rt_destroy(m_ctx);
    m_ctx = nullptr;
  }
  rt_ctx* ctx() const { return m_ctx; }
  const FlatScene& flat() const { return m_flat; }
  bool valid() const { return m_ctx != nullptr; }

  static void check(int rc, const char* what) {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
  }

 private:
  void open(int device) {
    rt_options opt = {};
    opt.device = device;
    check(rt_create(&m_flat.desc, &opt, &m_ctx), "rt_create");
  }
  FlatScene m_flat;
  rt_ctx* m_ctx;
};

// N devices behind one Renderer::render (rt_group: replicated scene, tile-sharded frame,
// owned tiles to devices[0] over xGMI)
class GpuGroupSession {
 public:
  GpuGroupSession(const Scene& scene, const std::vector<int>& devices) : m_flat(scene), m_group(nullptr) {
    std::vector<int32_t> d(devices.begin(), devices.end());
    GpuSession::check(rt_group_create(&m_flat.desc, d.data(), static_cast<uint32_t>(d.size()), nullptr, &m_group),
                      "rt_group_create");
  }
  ~GpuGroupSession() {
    if (m_group) rt_group_destroy(m_group);
  }
  GpuGroupSession(const GpuGroupSession&) = delete;
  GpuGroupSession& operator=(const GpuGroupSession&) = delete;
  rt_group* group() const { return m_group; }
  rt_ctx* ctx0() const { return rt_group_ctx(m_group, 0); }

 private:
  FlatScene m_flat;
  rt_group* m_group;
};

// process-wide settings the reference has no place for (set by Main.cpp's extra
// flags; defaults reproduce `./RayTracer` without them)
struct GpuSettings {
  int device = 0;
  unsigned seed = 1;
  unsigned accel = RT_ACCEL_BVH;
  unsigned progress = 0;  // samples per launch / per update.ppm (0 = whole frame at once)
  double tune = 0.;       // seconds of measured-cost BVH tuning before the frame (rt_bvh_tune; 0 = none)
  std::vector<int> devices;  // more than one entry: Renderer::render tile-shards the frame over them (rt_group)
  static GpuSettings& get() {
    static GpuSettings s;
    return s;
  }
};
