// ScenePresets.cpp — see ScenePresets.h.  Restates the scene constants of the
// reference application; validated bit-for-bit against a dump of the reference's
// own Scene object (tests/golden/ref_vectors.json: cubes_*, lowres_*).
#include "ScenePresets.h"

#include <cmath>
#include <stdexcept>

namespace rtpreset {

namespace {

struct Quad {
  float c[4][3];
  float n[3];
};

void appendQuad(Mesh& mesh, const Quad& q) {
  const int base = static_cast<int>(mesh.vertexPositions().size());
  const Vec3f normal(q.n[0], q.n[1], q.n[2]);
  for (int i = 0; i < 4; ++i) {
    mesh.vertexPositions().push_back(Vec3f(q.c[i][0], q.c[i][1], q.c[i][2]));
    mesh.vertexNormals().push_back(normal);
  }
  // two triangles sharing the 0-3 diagonal (Main.cpp:33-36)
  mesh.indexedTriangles().push_back(Triangle(base, base + 1, base + 3));
  mesh.indexedTriangles().push_back(Triangle(base, base + 2, base + 3));
}

// C5: one mesh of 83,334 scaled/translated copies of `unit` on a 44^3 lattice,
// lexicographic cell order, no randomness (SURVEY.md §8d "stress scene").
// ("stress8": the same on an 88^3 lattice, 666,672 copies = 8.0 M triangles — a supplementary
// point whose node + triangle arrays (0.5 GB) exceed the 256 MB Infinity Cache.)
Mesh latticeOfCopies(const Mesh& unit, int G, int copies) {
  const float lo[3] = {-1.4f, -0.95f, -1.4f}, hi[3] = {1.4f, 1.4f, 1.0f};
  Vec3f bmin = unit.vertexPositions()[0], bmax = bmin;
  for (const Vec3f& p : unit.vertexPositions())
    for (int a = 0; a < 3; ++a) {
      if (p[a] < bmin[a]) bmin[a] = p[a];
      if (p[a] > bmax[a]) bmax[a] = p[a];
    }
  float cell[3], ext = 0.f;
  for (int a = 0; a < 3; ++a) {
    cell[a] = (hi[a] - lo[a]) / static_cast<float>(G);
    if (bmax[a] - bmin[a] > ext) ext = bmax[a] - bmin[a];
  }
  Mesh out;
  out.material() = unit.material();
  const int nv = static_cast<int>(unit.vertexPositions().size());
  out.vertexPositions().reserve(static_cast<size_t>(copies) * nv);
  out.indexedTriangles().reserve(static_cast<size_t>(copies) * unit.indexedTriangles().size());
  for (int j = 0; j < copies; ++j) {
    const int ix = j / (G * G), iy = (j / G) % G, iz = j % G;
    const int cellIdx[3] = {ix, iy, iz};
    const int base = j * nv;
    for (const Vec3f& p : unit.vertexPositions()) {
      Vec3f q;
      for (int a = 0; a < 3; ++a) {
        const float s = 0.4f * cell[a] / ext;  // 0.4 x cell, aspect preserved per axis cell size
        const float centre = lo[a] + (static_cast<float>(cellIdx[a]) + 0.5f) * cell[a];
        q[a] = centre + (p[a] - 0.5f * (bmin[a] + bmax[a])) * s;
      }
      out.vertexPositions().push_back(q);
    }
    for (const Triangle& t : unit.indexedTriangles())
      out.indexedTriangles().push_back(Triangle(base + t[0], base + t[1], base + t[2]));
  }
  out.recomputeNormals();
  return out;
}

}  // namespace

void rotationY(Mesh& mesh, float phi) {
  const float c = std::cos(phi), s = std::sin(phi);  // float overloads, as Main.cpp:89-90
  const Vec3f r0(c, 0.f, s), r1(0.f, 1.f, 0.f), r2(-s, 0.f, c);
  for (Vec3f& p : mesh.vertexPositions()) p = Vec3f(dot(r0, p), dot(r1, p), dot(r2, p));
}

Scene buildCornellScene(const std::string& kind, const std::string& meshDir, size_t width, size_t height) {
  Scene scene;
  scene.camera() = Camera(Vec3f(0.3f, 0.6f, 2.3f), Vec3f(), Vec3f(0.f, 1.f, 0.f), 60.f,
                          static_cast<float>(width) / static_cast<float>(height));

  const Vec3f white(1.f, 1.f, 1.f);
  scene.lightsources().push_back(LightSource(Vec3f(-1.4f, 1.f, 2.9f), white, Vec3f(0.3f, 0.f, -1.f), 0.85f, 0.01f));
  scene.lightsources().push_back(LightSource(Vec3f(1.4f, 1.f, 2.9f), white, Vec3f(-0.3f, 0.f, -1.f), 0.85f, 0.01f));
  scene.lightsources().push_back(LightSource(Vec3f(0.f, -0.3f, 1.1f), white, Vec3f(0.f, 0.f, -1.f), 0.85f, 0.1f));

  Mesh walls, leftWall, rightWall, slot3, slot4;
  const Vec3f wallF0(0.5f, 0.5f, 0.5f);
  walls.material() = Material(0.6f, 0.3f, Vec3f(0.96f, 0.96f, 0.86f), wallF0);
  leftWall.material() = Material(0.6f, 0.3f, Vec3f(0.9f, 0.3f, 0.3f), wallF0);
  rightWall.material() = Material(0.6f, 0.3f, Vec3f(0.3f, 0.9f, 0.3f), wallF0);
  slot3.material() = Material(0.1f, 0.1f, Vec3f(0.9f, 0.9f, 0.9f), Vec3f(1.0f, 0.86f, 0.57f));
  slot4.material() = Material(0.8f, 0.9f, Vec3f(0.4f, 0.4f, 0.9f),
                              Vec3f(static_cast<float>(0.3), static_cast<float>(0.3), static_cast<float>(0.3)));

  std::string file3;
  bool lattice = false;
  if (kind == "cubes") file3 = "cube_tri.off";
  else if (kind == "lowres") file3 = "example_low_res.off";
  else if (kind == "hires") file3 = "example.off";
  else if (kind == "stress" || kind == "stress8") file3 = "cube_tri.off", lattice = true;
  else if (kind.rfind("file:", 0) == 0) file3 = kind.substr(5);
  else throw std::runtime_error("unknown scene kind '" + kind + "'");
  slot3.loadOFF(meshDir + "/" + file3);
  slot4.loadOFF(meshDir + "/cube_tri2.off");

  const float B = 1.51f, F = -1.f, C = 1.5f;  // half size, floor y, ceiling y
  appendQuad(walls, Quad{{{B, F, B}, {B, F, -B}, {-B, F, B}, {-B, F, -B}}, {0.f, 1.f, 0.f}});     // floor
  appendQuad(walls, Quad{{{-B, F, -B}, {B, F, -B}, {-B, C, -B}, {B, C, -B}}, {0.f, 0.f, 1.f}});   // back
  appendQuad(walls, Quad{{{B, C, B}, {B, C, -B}, {-B, C, B}, {-B, C, -B}}, {0.f, -1.f, 0.f}});    // ceiling
  appendQuad(leftWall, Quad{{{-B, F, B}, {-B, F, -B}, {-B, C, B}, {-B, C, -B}}, {1.f, 0.f, 0.f}});
  appendQuad(rightWall, Quad{{{B, F, B}, {B, F, -B}, {B, C, B}, {B, C, -B}}, {-1.f, 0.f, 0.f}});

  if (lattice) {
    slot3 = kind == "stress8" ? latticeOfCopies(slot3, 88, 666672) : latticeOfCopies(slot3, 44, 83334);
  } else {
    rotationY(slot3, static_cast<float>(3.14159265358979323846 / 4.5f));
  }
  rotationY(slot4, static_cast<float>(-3.14159265358979323846 / 4.5f));

  scene.meshes().push_back(walls);
  scene.meshes().push_back(leftWall);
  scene.meshes().push_back(rightWall);
  scene.meshes().push_back(slot3);
  scene.meshes().push_back(slot4);
  return scene;
}

}  // namespace rtpreset
