// Mesh.h — indexed triangle mesh container + OFF loader, host side.
// Public surface of reference source/Mesh.h:16-142 (vertexPositions /
// vertexNormals / indexedTriangles / material / loadOFF / recomputeNormals).
// Acceleration data is not kept per mesh: the GPU context builds ONE flattened
// BVH over the whole scene (csrc/bvh_build.cpp), so computeAABB()/computeBVH()
// are accepted and do nothing.
#pragma once

#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Material.h"
#include "Vec3.h"

typedef Vec3i Triangle;

class Mesh {
 public:
  Mesh() {}

  const std::vector<Vec3f>& vertexPositions() const { return m_vertexPositions; }
  std::vector<Vec3f>& vertexPositions() { return m_vertexPositions; }
  const std::vector<Vec3f>& vertexNormals() const { return m_vertexNormals; }
  std::vector<Vec3f>& vertexNormals() { return m_vertexNormals; }
  const std::vector<Triangle>& indexedTriangles() const { return m_indexedTriangles; }
  std::vector<Triangle>& indexedTriangles() { return m_indexedTriangles; }
  const Material& material() const { return m_material; }
  Material& material() { return m_material; }

  // Per-vertex normal = normalised sum of the unit normals of the incident
  // triangles (no area weighting), as reference Mesh.h:45-55.
  void recomputeNormals() {
    m_vertexNormals.resize(m_vertexPositions.size(), Vec3f());
    for (const Triangle& t : m_indexedTriangles) {
      const Vec3f& a = m_vertexPositions[t[0]];
      const Vec3f fn = normalize(cross(m_vertexPositions[t[1]] - a, m_vertexPositions[t[2]] - a));
      for (int c = 0; c < 3; ++c) m_vertexNormals[t[c]] += fn;
    }
    for (Vec3f& n : m_vertexNormals) n.normalize();
  }

  // Geomview OFF: header token, counts, vertices, then faces; polygons are fan
  // triangulated around their first vertex (reference Mesh.h:57-90).
  void loadOFF(const std::string& filename) {
    m_vertexPositions.clear();
    m_indexedTriangles.clear();
    std::ifstream in(filename.c_str());
    if (!in) throw std::runtime_error("Error loading OFF file: " + filename);
    std::string magic;
    unsigned nv = 0, nf = 0, ne = 0;
    in >> magic;
    skipComment(in);
    in >> nv >> nf >> ne;
    skipComment(in);
    m_vertexPositions.resize(nv);
    for (unsigned i = 0; i < nv; ++i) in >> m_vertexPositions[i];
    for (unsigned f = 0; f < nf; ++f) {
      unsigned corners = 0;
      in >> corners;
      std::vector<unsigned> idx(corners);
      for (unsigned c = 0; c < corners; ++c) in >> idx[c];
      for (unsigned c = 2; c < corners; ++c) m_indexedTriangles.push_back(Triangle(idx[0], idx[c - 1], idx[c]));
    }
    if (in.fail() && !in.eof()) throw std::runtime_error("Error Loading OFF file: malformed " + filename);
    recomputeNormals();
  }

  void computeAABB() {}
  void computeBVH() {}

 private:
  static void skipComment(std::ifstream& in) {
    while (in.peek() == '\n' || in.peek() == ' ' || in.peek() == '\r') in.get();
    if (in.peek() == '#') {
      std::string rest;
      std::getline(in, rest);
    }
  }

  std::vector<Vec3f> m_vertexPositions;
  std::vector<Vec3f> m_vertexNormals;
  std::vector<Triangle> m_indexedTriangles;
  Material m_material;
};
