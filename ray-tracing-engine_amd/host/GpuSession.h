// GpuSession.h — RAII owner of an rt_ctx for the host classes.  Every host class
// that has to EVALUATE something (Renderer, RayTracer, PhotonMap, kdtree) goes
// through one of these; construction throws std::runtime_error with the C ABI's
// message when the HIP library cannot reach a gfx950 device — there is no host
// implementation to fall back to.
#pragma once

#include <stdexcept>
#include <string>

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

// process-wide settings the reference has no place for (set by Main.cpp's extra
// flags; defaults reproduce `./RayTracer` without them)
struct GpuSettings {
  int device = 0;
  unsigned seed = 1;
  unsigned accel = RT_ACCEL_BVH;
  unsigned progress = 0;  // samples per launch / per update.ppm (0 = whole frame at once)
  static GpuSettings& get() {
    static GpuSettings s;
    return s;
  }
};
