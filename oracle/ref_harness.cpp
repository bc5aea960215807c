// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Driver that compiles the *reference's own sources where they lie* under
// /root/reference/source (via -I, nothing is copied) and exercises their public
// and — through the `private -> public` macro below — private interfaces, so
// that the restatement in oracle/rt_oracle.cpp and the HIP path can be pinned
// to the real thing.  Build recipe: oracle/Makefile (target _ref/ref_harness).
// The produced binary lives in oracle/_ref/ (git-ignored) and is only ever
// executed by tests/, bench.py's cpu_baseline leg and tests/golden/make_golden.py.
//
// Everything in this file is our own code; the scene script below restates the
// constants of reference source/Main.cpp:26-151,165-208 and
// is validated by reproducing the md5 of the stock binary's output (App. C).
//
// Sub-commands
//   render  <meshdir> <scene> <w> <h> <mode> <N> <p> <k> <out.ppm>
//   vectors <meshdir> <out.json>
//   time    <meshdir> <scene> <w> <h> <mode> <N> <p> <k>     (prints seconds)
#define _USE_MATH_DEFINES
#include <assert.h>
// NB: <math.h> must NOT be pre-included: libstdc++'s wrapper would pull the float
// overloads into :: and change what `tan(...)` in Camera.h:15 resolves to.

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

// open the reference classes up for white-box dumps (std headers above are
// already include-guarded, so the macro only touches the reference headers)
#define private public
#define protected public
// same include order as reference source/Main.cpp:13-22 (it decides which
// cmath overloads are visible where)
#include "Camera.h"
#include "CommandLine.h"
#include "Image.cpp"
#include "LightSource.h"
#include "Material.h"
#include "Mesh.h"
#include "Ray.h"
#include "RayTracer.h"
#include "Renderer.cpp"
#include "Scene.h"
#undef private
#undef protected

namespace {

uint32_t fbits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

// ---------------------------------------------------------------- scene script
void addQuad(Mesh& m, const Vec3f& a, const Vec3f& b, const Vec3f& c,
             const Vec3f& d, const Vec3f& n) {
  int base = (int)m.vertexPositions().size();
  for (const Vec3f& p : {a, b, c, d}) {
    m.vertexPositions().push_back(p);
    m.vertexNormals().push_back(n);
  }
  m.indexedTriangles().push_back(Vec3i(base, base + 1, base + 3));
  m.indexedTriangles().push_back(Vec3i(base, base + 2, base + 3));
}

void spinY(Mesh& m, float phi) {
  // Main.cpp:88-99 — positions only; cos/sin resolve to the float overloads
  // because `using namespace std` (CommandLine.h) is in force here as there.
  float c = cos(phi), s = sin(phi);
  Vec3f r0(c, 0, s), r1(0, 1, 0), r2(-s, 0, c);
  for (Vec3f& p : m.vertexPositions()) {
    Vec3f q(dot(r0, p), dot(r1, p), dot(r2, p));
    p = q;
  }
}

Scene buildScene(const std::string& meshdir, const std::string& kind, size_t w,
                 size_t h) {
  Scene scene;
  scene.camera() = Camera(Vec3f(0.3f, 0.6f, 2.3f), Vec3f(), Vec3f(0.f, 1.f, 0.f),
                          60.f, float(w) / h);
  scene.lightsources().push_back(LightSource(Vec3f(-1.4f, 1.f, 2.9f),
                                             Vec3f(1.f, 1.f, 1.f),
                                             Vec3f(0.3f, 0.f, -1.f), 0.85f, 0.01f));
  scene.lightsources().push_back(LightSource(Vec3f(1.4f, 1.f, 2.9f),
                                             Vec3f(1.f, 1.f, 1.f),
                                             Vec3f(-0.3f, 0.f, -1.f), 0.85f, 0.01f));
  scene.lightsources().push_back(LightSource(Vec3f(0.f, -0.3f, 1.1f),
                                             Vec3f(1.f, 1.f, 1.f),
                                             Vec3f(0.f, 0.f, -1.f), 0.85f, 0.1f));
  Mesh walls, left, right, slot3, slot4;
  Vec3f wallsF0(0.5f, 0.5f, 0.5f);
  walls.material() = Material(0.6f, 0.3f, Vec3f(0.96f, 0.96f, 0.86f), wallsF0);
  left.material() = Material(0.6f, 0.3f, Vec3f(0.9f, 0.3f, 0.3f), wallsF0);
  right.material() = Material(0.6f, 0.3f, Vec3f(0.3f, 0.9f, 0.3f), wallsF0);
  slot3.material() =
      Material(0.1f, 0.1f, Vec3f(0.9f, 0.9f, 0.9f), Vec3f(1.0f, 0.86f, 0.57f));
  slot4.material() =
      Material(0.8f, 0.9f, Vec3f(0.4f, 0.4f, 0.9f), Vec3f(0.3, 0.3, 0.3));

  std::string f3 = "cube_tri.off";
  if (kind == "lowres") f3 = "example_low_res.off";
  else if (kind == "hires") f3 = "example.off";
  else if (kind != "cubes") throw std::runtime_error("unknown scene " + kind);
  slot3.loadOFF(meshdir + "/" + f3);
  slot4.loadOFF(meshdir + "/cube_tri2.off");

  const float B = 1.51f, C = 1.5f;
  addQuad(walls, Vec3f(B, -1.f, B), Vec3f(B, -1.f, -B), Vec3f(-B, -1.f, B),
          Vec3f(-B, -1.f, -B), Vec3f(0.f, 1.f, 0.f));
  addQuad(walls, Vec3f(-B, -1.f, -B), Vec3f(B, -1.f, -B), Vec3f(-B, C, -B),
          Vec3f(B, C, -B), Vec3f(0.f, 0.f, 1.f));
  addQuad(walls, Vec3f(B, C, B), Vec3f(B, C, -B), Vec3f(-B, C, B),
          Vec3f(-B, C, -B), Vec3f(0.f, -1.f, 0.f));
  // left wall x=-B (Main.cpp:64-71), right wall x=+B (:73-81)
  addQuad(left, Vec3f(-B, -1.f, B), Vec3f(-B, -1.f, -B), Vec3f(-B, C, B),
          Vec3f(-B, C, -B), Vec3f(1.f, 0.f, 0.f));
  addQuad(right, Vec3f(B, -1.f, B), Vec3f(B, -1.f, -B), Vec3f(B, C, B),
          Vec3f(B, C, -B), Vec3f(-1.f, 0.f, 0.f));

  spinY(slot3, M_PI / 4.5f);
  spinY(slot4, -M_PI / 4.5f);
  scene.meshes().push_back(walls);
  scene.meshes().push_back(left);
  scene.meshes().push_back(right);
  scene.meshes().push_back(slot3);
  scene.meshes().push_back(slot4);
  return scene;
}

// ------------------------------------------------------------------ JSON bits
struct J {
  std::ostringstream o;
  bool first = true;
  void key(const char* k) {
    if (!first) o << ",\n";
    first = false;
    o << "\"" << k << "\":";
  }
  template <class It>
  void arr(It b, It e) {
    o << "[";
    for (It i = b; i != e; ++i) {
      if (i != b) o << ",";
      o << *i;
    }
    o << "]";
  }
  void u32(const char* k, const std::vector<uint32_t>& v) {
    key(k);
    arr(v.begin(), v.end());
  }
  void i64(const char* k, const std::vector<long long>& v) {
    key(k);
    arr(v.begin(), v.end());
  }
};

void push3(std::vector<uint32_t>& v, const Vec3f& p) {
  v.push_back(fbits(p[0]));
  v.push_back(fbits(p[1]));
  v.push_back(fbits(p[2]));
}

void dumpScene(J& j, const std::string& tag, const Scene& sc) {
  std::vector<uint32_t> pos, nrm, mat, tri, toff, voff;
  uint32_t to = 0, vo = 0;
  for (const Mesh& m : sc.meshes()) {
    toff.push_back(to);
    voff.push_back(vo);
    for (auto& p : m.vertexPositions()) push3(pos, p);
    for (auto& n : m.vertexNormals()) push3(nrm, n);
    for (auto& t : m.indexedTriangles()) {
      tri.push_back(t[0]);
      tri.push_back(t[1]);
      tri.push_back(t[2]);
    }
    to += m.indexedTriangles().size();
    vo += m.vertexPositions().size();
    const Material& M = m.material();
    mat.push_back(fbits(M.m_kd));
    mat.push_back(fbits(M.m_alpha));
    push3(mat, M.m_albedo);
    push3(mat, M.m_F0);
  }
  toff.push_back(to);
  voff.push_back(vo);
  j.u32((tag + "_pos").c_str(), pos);
  j.u32((tag + "_nrm").c_str(), nrm);
  j.u32((tag + "_tri_local").c_str(), tri);
  j.u32((tag + "_tri_off").c_str(), toff);
  j.u32((tag + "_vtx_off").c_str(), voff);
  j.u32((tag + "_mat").c_str(), mat);
  std::vector<uint32_t> cam, lights;
  const Camera& c = sc.camera();
  push3(cam, c.m_position);
  push3(cam, c.m_lowerLeftCorner);
  push3(cam, c.m_horizontal);
  push3(cam, c.m_vertical);
  j.u32((tag + "_cam").c_str(), cam);
  for (const LightSource& l : sc.lightsources()) {
    push3(lights, l.m_position);
    push3(lights, l.m_color);
    push3(lights, l.m_vertical);
    push3(lights, l.m_horizontal);
    push3(lights, l.m_normal);
    lights.push_back(fbits(l.m_intensity));
    lights.push_back(fbits(l.m_sideLength));
    lights.push_back(fbits(l.m_factor));
    lights.push_back(fbits(l.ac));
    lights.push_back(fbits(l.al));
    lights.push_back(fbits(l.aq));
  }
  j.u32((tag + "_lights").c_str(), lights);
}

// deterministic helper stream for building test inputs (NOT the engine `gen`)
struct Lcg {
  uint64_t s;
  explicit Lcg(uint64_t seed) : s(seed) {}
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
  float uni(float a, float b) { return a + (b - a) * (next() / 2147483648.0f); }
};

int cmdVectors(const std::string& meshdir, const std::string& out) {
  J j;
  j.o << "{\n";
  // ---- scenes (cubes in full; lowres in full too: 641 verts, small)
  Scene cubes = buildScene(meshdir, "cubes", 256, 256);
  Scene lowres = buildScene(meshdir, "lowres", 256, 256);
  dumpScene(j, "cubes", cubes);
  dumpScene(j, "lowres", lowres);
  {
    // non-square aspect camera frame
    Scene s = buildScene(meshdir, "cubes", 380, 270);
    std::vector<uint32_t> cam;
    push3(cam, s.camera().m_position);
    push3(cam, s.camera().m_lowerLeftCorner);
    push3(cam, s.camera().m_horizontal);
    push3(cam, s.camera().m_vertical);
    j.u32("cam_380x270", cam);
  }

  // ---- Camera::rayAt on a grid (Camera.h:27-30)
  {
    std::vector<uint32_t> v;
    for (int a = 0; a <= 8; a++)
      for (int b = 0; b <= 8; b++) {
        float u = a / 8.f, w = b / 8.f;
        Ray r = cubes.camera().rayAt(u, w);
        v.push_back(fbits(u));
        v.push_back(fbits(w));
        push3(v, r.origin());
        push3(v, r.direction());
      }
    j.u32("rayAt", v);
  }

  // ---- Ray::triangleIntersect (Ray.cpp:9-24): random, grazing, degenerate
  {
    std::vector<uint32_t> v;
    Lcg g(12345);
    for (int i = 0; i < 600; i++) {
      Vec3f p0(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      Vec3f p1(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      Vec3f p2(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      Vec3f o(g.uni(-2, 2), g.uni(-2, 2), g.uni(-2, 2));
      Vec3f d;
      int kind = i % 6;
      if (kind == 0) {  // aim at the interior
        float a = g.uni(0, 1), b = g.uni(0, 1 - a);
        d = (p0 * (1 - a - b) + p1 * a + p2 * b) - o;
      } else if (kind == 1) {  // aim exactly at an edge point
        float a = g.uni(0, 1);
        d = (p0 * (1 - a) + p1 * a) - o;
      } else if (kind == 2) {  // aim exactly at a vertex
        d = p2 - o;
      } else if (kind == 3) {  // origin on the triangle (self-hit acne case)
        float a = g.uni(0, 1), b = g.uni(0, 1 - a);
        o = p0 * (1 - a - b) + p1 * a + p2 * b;
        d = Vec3f(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      } else if (kind == 4) {  // nearly parallel to the plane
        Vec3f e = normalize(p1 - p0);
        d = e + normalize(cross(p1 - p0, p2 - p0)) * g.uni(-2e-6f, 2e-6f);
      } else {
        d = Vec3f(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      }
      if (i % 50 == 49) p2 = p1;  // degenerate triangle
      if (i % 2) d = normalize(d);
      float u = -7.f, w = -7.f, t = -7.f;  // sentinel: outputs may stay unwritten
      bool hit = Ray(o, d).triangleIntersect(p0, p1, p2, u, w, t);
      push3(v, p0);
      push3(v, p1);
      push3(v, p2);
      push3(v, o);
      push3(v, d);
      v.push_back(hit ? 1u : 0u);
      v.push_back(fbits(u));
      v.push_back(fbits(w));
      v.push_back(fbits(t));
    }
    j.u32("triangleIntersect", v);
  }

  // ---- RayTracer::rayTrace (RayTracer.h:27-53): camera, bounce and shadow rays
  auto traceSet = [&](const char* name, const Scene& sc, int n) {
    std::vector<uint32_t> v;
    RayTracer rt;
    Lcg g(777);
    for (int i = 0; i < n; i++) {
      Ray r = sc.camera().rayAt(g.uni(0, 1), g.uni(0, 1));
      for (int hop = 0; hop < 3; hop++) {
        size_t mi = 9999;
        Vec3i tri(-1, -1, -1);
        float u = -7.f, w = -7.f, d = -7.f;
        bool hit = rt.rayTrace(r, sc, mi, tri, u, w, d);
        push3(v, r.origin());
        push3(v, r.direction());
        v.push_back(hit);
        v.push_back((uint32_t)mi);
        v.push_back(tri[0]);
        v.push_back(tri[1]);
        v.push_back(tri[2]);
        v.push_back(fbits(u));
        v.push_back(fbits(w));
        v.push_back(fbits(d));
        if (!hit) break;
        const Mesh& m = sc.meshes()[mi];
        float ww = 1.f - u - w;
        Vec3f p = ww * m.vertexPositions()[tri[0]] + u * m.vertexPositions()[tri[1]] +
                  w * m.vertexPositions()[tri[2]];
        // next ray starts exactly on the surface (no epsilon) like the reference
        Vec3f nd;
        if (hop == 0)
          nd = Vec3f(0.f, -0.3f, 1.1f) - p;  // un-normalised, towards light 2
        else
          nd = normalize(Vec3f(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1)));
        r = Ray(p, nd);
      }
    }
    j.u32(name, v);
  };
  traceSet("rayTrace_cubes", cubes, 400);
  traceSet("rayTrace_lowres", lowres, 250);

  // ---- Material::evaluateColorResponse (Material.h:25-70)
  {
    std::vector<uint32_t> v;
    Lcg g(4242);
    for (const Mesh& m : cubes.meshes()) {
      Material M = m.material();
      for (int i = 0; i < 60; i++) {
        Vec3f n(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
        Vec3f wi(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
        Vec3f wo(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
        if (i % 10 == 7) wo = wi * -1.f;            // wh = 0
        if (i % 10 == 8) wi = cross(n, wo);         // grazing n.wi = 0
        if (i % 10 == 9) n = Vec3f(0, 1, 0), wi = Vec3f(0.3f, 0.8f, 0.1f), wo = Vec3f(-0.2f, 0.5f, 0.4f);
        Vec3f r = M.evaluateColorResponse(n, wi, wo);
        push3(v, n);
        push3(v, wi);
        push3(v, wo);
        v.push_back(fbits(M.m_kd));
        v.push_back(fbits(M.m_alpha));
        push3(v, M.m_albedo);
        push3(v, M.m_F0);
        push3(v, r);
      }
    }
    j.u32("bsdf", v);
  }

  // ---- LightSource::evaluateLight (LightSource.h:51-59)
  {
    std::vector<uint32_t> v;
    Lcg g(99);
    for (LightSource l : cubes.lightsources())
      for (int i = 0; i < 20; i++) {
        Vec3f p(g.uni(-1.5f, 1.5f), g.uni(-1.f, 1.5f), g.uni(-1.5f, 1.5f));
        push3(v, p);
        push3(v, l.evaluateLight(p));
      }
    j.u32("evaluateLight", v);
  }

  // ---- the engine and the three samplers, from a fresh seed-1 engine
  {
    gen.seed(1);
    std::vector<uint32_t> raw;
    for (int i = 0; i < 8; i++) raw.push_back((uint32_t)gen());
    j.u32("engine_first8", raw);

    gen.seed(1);
    RayTracer rt;
    std::vector<uint32_t> v;
    for (int N : {1, 2, 4, 8, 16, 128})
      for (int i = 0; i < std::min(N, 12); i++) {
        Vec3f s = rt.jitterSample(i, N);
        v.push_back(N);
        v.push_back(i);
        v.push_back(fbits(s[0]));
        v.push_back(fbits(s[1]));
      }
    j.u32("jitterSample_seq", v);

    gen.seed(1);
    std::vector<uint32_t> a;
    for (int rep = 0; rep < 4; rep++)
      for (LightSource l : cubes.lightsources()) push3(a, l.randAreaPosition());
    j.u32("randAreaPosition_seq", a);

    gen.seed(1);
    std::vector<uint32_t> hs;
    Lcg g(31337);
    for (int i = 0; i < 200; i++) {
      Vec3f n(g.uni(-1, 1), g.uni(-1, 1), g.uni(-1, 1));
      if (i % 7 == 0) n = Vec3f(0, 1, 0);
      if (i % 7 == 1) n = Vec3f(0, 0, -1);
      if (i % 7 == 2) n = Vec3f(1, 0, 0);
      Vec3f d = rt.hsphereUniformSample(n, M_PI / 2.f);
      push3(hs, n);
      push3(hs, d);
    }
    j.u32("hsphere_seq", hs);
  }

  // ---- photon map (PhotonMap.h:14-50,92-155) + kd-tree order + knearest
  {
    gen.seed(1);
    RayTracer rt;
    PhotonMap pm(cubes, 3000, rt);
    std::vector<uint32_t> ph;
    for (const Particle& p : pm.list()) {
      push3(ph, p.position());
      push3(ph, p.incomeDirection());
      ph.push_back(fbits(p.weight()));
    }
    j.u32("photons_cubes_3000", ph);
    std::vector<uint32_t> after;
    for (int i = 0; i < 4; i++) after.push_back((uint32_t)gen());
    j.u32("photons_cubes_3000_engine_after", after);

    kdtree tree(pm.list().begin(), pm.list().end());
    std::vector<uint32_t> order;  // node array order after make_tree
    for (auto& n : tree.m_nodes) push3(order, n.m_point.position());
    j.u32("kdtree_order_pos", order);

    std::vector<uint32_t> q;
    Lcg g(2024);
    for (int i = 0; i < 300; i++) {
      Particle t;
      if (i % 3 == 0) {
        const Particle& s = pm.list()[g.next() % pm.list().size()];
        t.position() = s.position() + Vec3f(g.uni(-0.05f, 0.05f), g.uni(-0.05f, 0.05f), g.uni(-0.05f, 0.05f));
      } else if (i % 3 == 1) {
        t.position() = Vec3f(g.uni(-1.5f, 1.5f), -1.f, g.uni(-1.5f, 1.5f));
      } else {
        t.position() = Vec3f(g.uni(-1.5f, 1.5f), g.uni(-1.f, 1.5f), g.uni(-1.5f, 1.5f));
      }
      if (i == 299) t.position() = pm.list()[5].position();  // exact hit: bestdist==0 path
      for (int k : {1, 5, 10}) {
        std::vector<Particle> res;
        tree.knearest(t, k, res);
        push3(q, t.position());
        q.push_back(k);
        q.push_back((uint32_t)tree.m_visited);
        for (auto& r : res) {
          push3(q, r.position());
          push3(q, r.incomeDirection());
        }
      }
    }
    j.u32("knearest", q);
  }

  j.o << "\n}\n";
  std::ofstream f(out);
  f << j.o.str();
  return 0;
}

int cmdRender(int argc, char** argv, bool timeOnly) {
  std::string meshdir = argv[2], kind = argv[3];
  size_t w = atoi(argv[4]), h = atoi(argv[5]);
  int mode = atoi(argv[6]), N = atoi(argv[7]), p = atoi(argv[8]), k = atoi(argv[9]);
  std::cout.setstate(std::ios::failbit);  // silence the progress bar
  Scene scene = buildScene(meshdir, kind, w, h);
  Image image(w, h);
  RayTracer rt;
  Renderer renderer;
  if (p > 0)
    renderer = Renderer(scene, N, mode, rt, p, k);
  else
    renderer = Renderer(scene, N, mode, rt);
  image.fillBackground();
  auto t0 = std::chrono::steady_clock::now();
  renderer.render(image);  // writes update.ppm into CWD every pass (Renderer.cpp:268)
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout.clear();
  if (!timeOnly) image.savePPM(argv[10]);
  printf("{\"seconds\": %.6f, \"samples\": %zu}\n", s, w * h * (size_t)N);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "vectors")) return cmdVectors(argv[2], argv[3]);
  if (argc >= 11 && !strcmp(argv[1], "render")) return cmdRender(argc, argv, false);
  if (argc >= 10 && !strcmp(argv[1], "time")) return cmdRender(argc, argv, true);
  fprintf(stderr, "usage: see header of oracle/ref_harness.cpp\n");
  return 2;
}
