// HostBindings.cpp — out-of-line pieces of the host layer and the C view of it
// (include/rt_host.h).  Pure host code; the only arithmetic done here is scene
// preparation the reference also does on the host (Main.cpp, Mesh.h, Image.cpp,
// kdtree.h:60-69).  Anything that evaluates the hot path forwards to the device
// through include/rt_amd.h and fails loudly when no GPU is present.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "Image.h"
#include "Ray.h"
#include "SceneFlatten.h"
#include "ScenePresets.h"
#include "rt_amd.h"
#include "rt_host.h"

// the reference's process-wide engine (LightSource.h:6), default-seeded (seed 1)
std::default_random_engine gen;

namespace {
thread_local std::string g_hostErr;

void unitOrThrow(uint32_t which, const void* in, void* out) {
  if (rt_test_unit(0, which, in, out, 1) != RT_OK)
    throw std::runtime_error(std::string("GPU evaluation failed: ") + rt_last_error());
}
}  // namespace

bool Ray::triangleIntersect(const Vec3f& p0, const Vec3f& p1, const Vec3f& p2, float& u, float& v,
                            float& t) const {
  float in[15], out[4] = {0.f, u, v, t};
  const Vec3f* src[5] = {&p0, &p1, &p2, &m_origin, &m_direction};
  for (int k = 0; k < 5; ++k)
    for (int c = 0; c < 3; ++c) in[3 * k + c] = (*src[k])[c];
  unitOrThrow(RT_UNIT_TRIANGLE, in, out);
  u = out[1], v = out[2], t = out[3];
  return out[0] != 0.f;
}

Vec3f Material::evaluateColorResponse(const Vec3f& normal, const Vec3f& wi, const Vec3f& wo) const {
  float in[17] = {m_kd, m_alpha, m_albedo[0], m_albedo[1], m_albedo[2], m_F0[0], m_F0[1], m_F0[2]};
  for (int c = 0; c < 3; ++c) in[8 + c] = normal[c], in[11 + c] = wi[c], in[14 + c] = wo[c];
  float out[3];
  unitOrThrow(RT_UNIT_BSDF, in, out);
  return Vec3f(out[0], out[1], out[2]);
}

struct rt_host_scene {
  FlatScene flat;
};

extern "C" {

const char* rt_host_last_error(void) { return g_hostErr.c_str(); }

int rt_host_scene_build(const char* kind, const char* mesh_dir, uint32_t width, uint32_t height,
                        rt_host_scene** out) {
  if (!kind || !mesh_dir || !out || width == 0 || height == 0) {
    g_hostErr = "invalid argument";
    return RT_ERR_INVALID;
  }
  *out = nullptr;
  try {
    Scene scene = rtpreset::buildCornellScene(kind, mesh_dir, width, height);
    rt_host_scene* s = new rt_host_scene();
    s->flat.assign(scene);
    *out = s;
    return RT_OK;
  } catch (const std::exception& e) {
    g_hostErr = e.what();
    return RT_ERR_INVALID;
  }
}

const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* s) { return s ? &s->flat.desc : nullptr; }
void rt_host_scene_free(rt_host_scene* s) { delete s; }

void rt_host_fill_background(float* rgb, uint32_t width, uint32_t height) {
  Image img(width, height);
  img.fillBackground();
  std::memcpy(rgb, img.data(), sizeof(float) * 3 * (size_t)width * height);
}

int rt_host_save_ppm(const char* path, const float* rgb, uint32_t width, uint32_t height) {
  Image img(width, height);
  std::memcpy(img.data(), rgb, sizeof(float) * 3 * (size_t)width * height);
  std::ofstream probe(path);
  if (!probe) {
    g_hostErr = std::string("cannot open ") + path;
    return RT_ERR_INVALID;
  }
  probe.close();
  img.savePPM(path);
  return RT_OK;
}

// kdtree::make_tree (kdtree.h:60-69): the node ARRAY ORDER after the recursive
// std::nth_element calls is the tree (node of [b,e) sits at b + (e-b)/2).  The
// approximate k-NN the reference performs depends on that exact order, so the
// same standard-library selection runs here on a key/index pair array.
int rt_host_kd_order(float* pos3, float* dir3, float* weight, uint32_t n) {
  if (n && (!pos3 || !dir3)) {
    g_hostErr = "null photon arrays";
    return RT_ERR_INVALID;
  }
  struct Item {
    float p[3];
    uint32_t src;
  };
  std::vector<Item> items(n);
  for (uint32_t i = 0; i < n; ++i) {
    std::memcpy(items[i].p, pos3 + 3 * (size_t)i, 12);
    items[i].src = i;
  }
  struct Range {
    size_t b, e, axis;
  };
  std::vector<Range> todo;
  todo.push_back({0, n, 0});
  while (!todo.empty()) {  // pre-order (left before right), as the recursion does
    const Range r = todo.back();
    todo.pop_back();
    if (r.e <= r.b) continue;
    const size_t mid = r.b + (r.e - r.b) / 2, ax = r.axis;
    std::nth_element(items.begin() + r.b, items.begin() + mid, items.begin() + r.e,
                     [ax](const Item& a, const Item& b) { return a.p[ax] < b.p[ax]; });
    const size_t next = (ax + 1) % 3;
    todo.push_back({mid + 1, r.e, next});
    todo.push_back({r.b, mid, next});
  }
  std::vector<float> np(3 * (size_t)n), nd(3 * (size_t)n), nw(weight ? n : 0);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t s = items[i].src;
    std::memcpy(&np[3 * (size_t)i], pos3 + 3 * (size_t)s, 12);
    std::memcpy(&nd[3 * (size_t)i], dir3 + 3 * (size_t)s, 12);
    if (weight) nw[i] = weight[s];
  }
  std::memcpy(pos3, np.data(), np.size() * 4);
  std::memcpy(dir3, nd.data(), nd.size() * 4);
  if (weight) std::memcpy(weight, nw.data(), nw.size() * 4);
  return RT_OK;
}

}  // extern "C"
