// SceneFlatten.h — Scene (vectors of meshes of vectors) -> rt_scene_desc.
//
// The GPU context wants one immutable snapshot: vertices of all meshes
// concatenated, triangles in the reference's (mesh, triangle) iteration order
// (reference source/RayTracer.h:32-35 — that order is the tie-break for equal
// hit distances), one material per mesh, the light table with its precomputed
// basis, and the camera frame.  FlatScene owns the arrays the descriptor points to.
#pragma once

#include <cstdint>
#include <vector>

#include "Scene.h"
#include "rt_amd.h"

struct FlatScene {
  std::vector<float> pos, nrm;
  std::vector<uint32_t> tri, triBegin, vtxBegin;
  std::vector<rt_material> materials;
  std::vector<rt_light> lights;
  rt_scene_desc desc;

  FlatScene() : desc() {}
  explicit FlatScene(const Scene& scene) : desc() { assign(scene); }
  FlatScene(const FlatScene&) = delete;
  FlatScene& operator=(const FlatScene&) = delete;

  static void put3(float* dst, const Vec3f& v) { dst[0] = v[0], dst[1] = v[1], dst[2] = v[2]; }

  void assign(const Scene& scene) {
    pos.clear(), nrm.clear(), tri.clear(), triBegin.clear(), vtxBegin.clear();
    materials.clear(), lights.clear();
    uint32_t vbase = 0, tbase = 0;
    for (const Mesh& m : scene.meshes()) {
      triBegin.push_back(tbase);
      vtxBegin.push_back(vbase);
      const size_t nv = m.vertexPositions().size();
      for (size_t i = 0; i < nv; ++i) {
        const Vec3f& p = m.vertexPositions()[i];
        // a mesh without normals (never the case after loadOFF) shades with null normals
        const Vec3f n = i < m.vertexNormals().size() ? m.vertexNormals()[i] : Vec3f();
        pos.insert(pos.end(), {p[0], p[1], p[2]});
        nrm.insert(nrm.end(), {n[0], n[1], n[2]});
      }
      for (const Triangle& t : m.indexedTriangles())
        for (int c = 0; c < 3; ++c) tri.push_back(vbase + static_cast<uint32_t>(t[c]));
      rt_material fm;
      fm.kd = m.material().kd(), fm.alpha = m.material().alpha();
      put3(fm.albedo, m.material().albedo());
      put3(fm.f0, m.material().F0());
      materials.push_back(fm);
      vbase += static_cast<uint32_t>(nv);
      tbase += static_cast<uint32_t>(m.indexedTriangles().size());
    }
    triBegin.push_back(tbase);
    vtxBegin.push_back(vbase);
    for (const LightSource& l : scene.lightsources()) {
      rt_light fl;
      put3(fl.position, l.position()), put3(fl.color, l.color());
      put3(fl.vertical, l.vertical()), put3(fl.horizontal, l.horizontal());
      put3(fl.normal, l.normal());
      fl.intensity = l.intensity(), fl.side = l.sideLength(), fl.factor = l.factor();
      fl.ac = l.attConstant(), fl.al = l.attLinear(), fl.aq = l.attQuadratic();
      lights.push_back(fl);
    }
    const Camera& c = scene.camera();
    put3(desc.camera.position, c.position()), put3(desc.camera.lower_left, c.lowerLeftCorner());
    put3(desc.camera.horizontal, c.horizontal()), put3(desc.camera.vertical, c.vertical());
    desc.n_meshes = static_cast<uint32_t>(scene.meshes().size());
    desc.n_vertices = vbase;
    desc.n_triangles = tbase;
    desc.n_lights = static_cast<uint32_t>(lights.size());
    desc.vertex_pos = pos.data(), desc.vertex_nrm = nrm.data();
    desc.tri_vtx = tri.data();
    desc.mesh_tri_begin = triBegin.data(), desc.mesh_vtx_begin = vtxBegin.data();
    desc.materials = materials.data(), desc.lights = lights.data();
  }
};
