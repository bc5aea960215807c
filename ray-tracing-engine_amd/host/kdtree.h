// kdtree.h — photon kd-tree, host side (API of reference source/kdtree.h:13-196).
//
// The tree is the ARRAY ORDER produced by the recursive std::nth_element of the
// reference's make_tree (:60-69): node of range [b,e) sits at b + (e-b)/2, left =
// [b,mid), right = (mid,e), axis cycles x,y,z.  No pointers are stored; the GPU
// k-NN kernel walks the same implicit tree.  knearest() runs ON THE GPU
// (rt_knn) and returns what the reference's routine returns, including its
// quirks (approximate result).
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "GpuSession.h"
#include "Particle.h"
#include "rt_host.h"

class kdtree {
 public:
  kdtree(const kdtree&) = delete;
  kdtree& operator=(const kdtree&) = delete;

  template <typename iterator>
  kdtree(iterator begin, iterator end) : m_visited(0) {
    for (iterator i = begin; i != end; ++i) m_nodes.push_back(*i);
    order();
  }
  template <typename func>
  kdtree(func&& f, size_t n) : m_visited(0) {
    m_nodes.reserve(n);
    for (size_t i = 0; i < n; ++i) m_nodes.push_back(f());
    order();
  }

  bool empty() const { return m_nodes.empty(); }
  size_t size() const { return m_nodes.size(); }
  size_t visited() const { return m_visited; }
  // photons in tree order (what rt_set_photons wants)
  const std::vector<Particle>& nodes() const { return m_nodes; }

  void knearest(const Particle& pt, int k, std::vector<Particle>& result) {
    if (m_nodes.empty()) throw std::logic_error("tree is empty");
    if (k > static_cast<int>(m_nodes.size())) throw std::logic_error("k is greater than the number of nodes");
    session();
    const float q[3] = {pt.position()[0], pt.position()[1], pt.position()[2]};
    std::vector<uint32_t> idx(k);
    std::vector<float> dist(k);
    uint32_t vis = 0;
    GpuSession::check(rt_knn(m_session->ctx(), q, 1, static_cast<uint32_t>(k), idx.data(), dist.data(), &vis), "rt_knn");
    m_visited = vis;
    for (int i = 0; i < k; ++i) result.push_back(m_nodes[idx[i]]);
  }

 private:
  void order() {
    const size_t n = m_nodes.size();
    std::vector<float> pos(3 * n), dir(3 * n), w(n);
    for (size_t i = 0; i < n; ++i) {
      for (int c = 0; c < 3; ++c) pos[3 * i + c] = m_nodes[i].position()[c], dir[3 * i + c] = m_nodes[i].incomeDirection()[c];
      w[i] = m_nodes[i].weight();
    }
    if (rt_host_kd_order(pos.data(), dir.data(), w.data(), static_cast<uint32_t>(n)) != RT_OK)
      throw std::runtime_error(rt_host_last_error());
    for (size_t i = 0; i < n; ++i)
      m_nodes[i] = Particle(Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                            Vec3f(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), w[i]);
  }
  // a stand-alone tree needs a device context only to hold the photon arrays
  void session() {
    if (m_session) return;
    Scene s;
    Mesh m;
    m.vertexPositions() = {Vec3f(0.f, 0.f, 0.f), Vec3f(1.f, 0.f, 0.f), Vec3f(0.f, 1.f, 0.f)};
    m.vertexNormals() = {Vec3f(0.f, 0.f, 1.f), Vec3f(0.f, 0.f, 1.f), Vec3f(0.f, 0.f, 1.f)};
    m.indexedTriangles().push_back(Triangle(0, 1, 2));
    s.meshes().push_back(m);
    m_session.reset(new GpuSession(s, GpuSettings::get().device));
    std::vector<float> pos(3 * m_nodes.size()), dir(3 * m_nodes.size());
    for (size_t i = 0; i < m_nodes.size(); ++i)
      for (int c = 0; c < 3; ++c) pos[3 * i + c] = m_nodes[i].position()[c], dir[3 * i + c] = m_nodes[i].incomeDirection()[c];
    GpuSession::check(rt_set_photons(m_session->ctx(), pos.data(), dir.data(), static_cast<uint32_t>(m_nodes.size())),
                      "rt_set_photons");
  }

  std::vector<Particle> m_nodes;
  size_t m_visited;
  std::unique_ptr<GpuSession> m_session;
};
