// bvh_build.cpp — binned-SAH top-down build (fork-join over the top levels), see bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace rtbvh {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; ++a) lo[a] = std::numeric_limits<float>::infinity(), hi[a] = -lo[a];
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], p[a]), hi[a] = std::max(hi[a], p[a]);
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], b.lo[a]), hi[a] = std::max(hi[a], b.hi[a]);
  }
  float halfArea() const {
    float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    if (d[0] < 0.f) return 0.f;
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
  }
};

struct Prim {
  Box box;
  float c[4];  // centroid; [3] = the SIZE key, -log2(longest box edge): a fourth "axis" to sweep or bin along
  uint32_t id;
};

// The primitive arrays without value-initialisation: a std::vector would zero 44 MB per million triangles on ONE thread (page
// faults included) before the threads that fill it get to touch it.
struct PrimBuf {
  std::unique_ptr<Prim[]> p;
  size_t n = 0;
  void resize(size_t k) { p.reset(new Prim[k]), n = k; }
  size_t size() const { return n; }
  Prim* begin() { return p.get(); }
  const Prim* begin() const { return p.get(); }
  Prim& operator[](size_t i) { return p[i]; }
  const Prim& operator[](size_t i) const { return p[i]; }
};

constexpr uint32_t kSweepMax = 4096;
TriRec makeRec(const rt_scene_desc& sc, uint32_t t, uint32_t mesh);  // 32768 built a slightly worse C4 tree (10.3 vs 10.0 nodes/ray)

struct Builder {
  const rt_scene_desc& sc;
  PrimBuf prims;
  Built& out;
  uint32_t leafMax;
  int depthCap = kMaxDepth - 1;  // deepest leaf level the tree may use
  uint32_t grain = ~0u;          // ranges above this many primitives fork a thread
  // threads a SINGLE big split may use for its passes over the range (the top levels of a big scene are otherwise one
  // thread walking a million primitives four times per level: 47 of the hybrid builder's 64 host milliseconds)
  uint32_t parThreads = 1;
  PrimBuf scratch;  // the parallel partition's second buffer (ranges are disjoint: concurrent splits do not collide)
  static constexpr uint32_t kParSplit = 131072u;
  // f(thread, begin, end) over contiguous chunks of [b, e); the chunking — hence any result that is merged in chunk
  // order — does not depend on how many threads actually ran
  template <class F>
  void parChunks(uint32_t b, uint32_t e, F f) const {
    const uint32_t T = std::max<uint32_t>(1u, std::min<uint32_t>(parThreads, (e - b) / 32768u));
    if (T <= 1u) {
      f(0u, b, e);
      return;
    }
    const uint64_t n = e - b;
    std::vector<std::thread> th;
    th.reserve(T - 1u);
    for (uint32_t t = 1; t < T; ++t) th.emplace_back([&f, b, n, t, T] { f(t, b + (uint32_t)(n * t / T), b + (uint32_t)(n * (t + 1) / T)); });
    f(0u, b, b + (uint32_t)(n / T));
    for (std::thread& x : th) x.join();
  }
  // primitives (boxes, centroids, size keys) of every triangle, validated; optionally the reference-order records
  float loadPrims(TriRec* trisRef) {
    prims.resize(sc.n_triangles);
    float maxAbs = 0.f;
    for (uint32_t m = 0; m < sc.n_meshes; ++m) {
      if (sc.mesh_tri_begin[m] > sc.mesh_tri_begin[m + 1]) throw std::runtime_error("mesh_tri_begin not monotone");
      std::vector<float> mx(64, 0.f);
      std::vector<int> bad(64, 0);
      parChunks(sc.mesh_tri_begin[m], sc.mesh_tri_begin[m + 1], [&](uint32_t th, uint32_t tb, uint32_t te) {
        float localMax = 0.f;
        for (uint32_t t = tb; t < te; ++t) {
          Prim& p = prims[t];
          p.id = t;
          p.box.reset();
          for (int k = 0; k < 3; ++k) {
            const uint32_t v = sc.tri_vtx[3 * static_cast<size_t>(t) + k];
            if (v < sc.mesh_vtx_begin[m] || v >= sc.mesh_vtx_begin[m + 1]) {
              bad[th] = 1;
              return;
            }
            const float* q = sc.vertex_pos + 3 * static_cast<size_t>(v);
            for (int a = 0; a < 3; ++a) {
              if (!std::isfinite(q[a])) {
                bad[th] = 2;
                return;
              }
              localMax = std::max(localMax, std::fabs(q[a]));
            }
            p.box.grow(q);
          }
          for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
          p.c[3] = -std::log2(std::max({p.box.hi[0] - p.box.lo[0], p.box.hi[1] - p.box.lo[1], p.box.hi[2] - p.box.lo[2], 1e-30f}));
          if (trisRef) trisRef[t] = makeRec(sc, t, m);
        }
        mx[th] = localMax;
      });
      for (int t = 0; t < 64; ++t) {
        if (bad[t] == 1) throw std::runtime_error("triangle references a vertex outside its mesh");
        if (bad[t] == 2) throw std::runtime_error("non-finite vertex position");
        maxAbs = std::max(maxAbs, mx[t]);
      }
    }
    return maxAbs;
  }
  // buildTop: ranges of at most this many primitives are not split here (0 = build the whole tree)
  uint32_t cutoff = 0;
  std::vector<TopBuilt::Part> parts;
  std::mutex partsLock;

  Builder(const rt_scene_desc& s, Built& o, uint32_t lm) : sc(s), out(o), leafMax(lm) {}

  Box boundsOf(uint32_t b, uint32_t e) const {
    Box r;
    r.reset();
    for (uint32_t i = b; i < e; ++i) r.grow(prims[i].box);
    return r;
  }

  // levels needed to split n primitives into leaves of <= leafMax by halving
  int levelsFor(uint32_t n) const {
    int l = 0;
    while (n > leafMax) n = (n + 1) / 2, ++l;
    return l;
  }

  // Chooses a split of [b,e) and partitions prims; returns the middle index.
  uint32_t split(uint32_t b, uint32_t e, int depth) {
    const uint32_t n = e - b;
    const uint32_t median = b + n / 2;
    const bool par = parThreads > 1u && n >= kParSplit;
    Box cb;
    cb.reset();
    float sLo = std::numeric_limits<float>::infinity(), sHi = -sLo;
    if (par) {  // one pass, every thread its chunk: centroid bounds and the size keys' range
      Box cbt[64];
      float lo4[64], hi4[64];
      for (int t = 0; t < 64; ++t) cbt[t].reset(), lo4[t] = sLo, hi4[t] = sHi;
      parChunks(b, e, [&](uint32_t t, uint32_t cb0, uint32_t ce0) {
        Box x;
        x.reset();
        float l = std::numeric_limits<float>::infinity(), h = -l;
        for (uint32_t i = cb0; i < ce0; ++i) x.grow(prims[i].c), l = std::min(l, prims[i].c[3]), h = std::max(h, prims[i].c[3]);
        cbt[t] = x, lo4[t] = l, hi4[t] = h;
      });
      for (int t = 0; t < 64; ++t) cb.grow(cbt[t]), sLo = std::min(sLo, lo4[t]), sHi = std::max(sHi, hi4[t]);
    } else {
      for (uint32_t i = b; i < e; ++i) cb.grow(prims[i].c);
    }
    // The size axis: candidates "the k largest primitives | the rest".  A wall quad among a mesh's triangles has its
    // centroid somewhere in the middle of the room, and every centroid split leaves it in a box with half the mesh;
    // sorted by size it comes off first (the device builder's size classes, bvh_gpu.hip, had found the same thing).
    // Node visits per ray / frame: C2 8.64 -> 7.80 / 50.1 -> 47.4 ms, C4 10.12 -> 8.55 / 937 -> 895 ms.  Greedy SAH is
    // not monotone in its candidate set: taken whenever it is cheapest, the size cut costs the 1 M-triangle lattice
    // 2.7 % (3.47 -> 3.57 triangle tests per ray) — so it has to beat the best centroid cut by a factor (1.5: the
    // wall-against-mesh cuts are that decisive; 1.2 and 1.5 give the same C2 / C4 trees, 310 / 312 / 311 ms on C5
    // for off / 1.2 / 1.5; profiles/r03_size_axis_ab.txt).  RT_BVH_SIZEAXIS: bit 0 swept ranges, bit 1 binned ranges.
    static const int sizeAxisMode = getenv("RT_BVH_SIZEAXIS") ? atoi(getenv("RT_BVH_SIZEAXIS")) : 3;
    static const float sizeBias = getenv("RT_BVH_SIZEBIAS") ? (float)atof(getenv("RT_BVH_SIZEBIAS")) : 1.5f;
    const bool sizeAxis = (sizeAxisMode & (n <= kSweepMax ? 1 : 2)) != 0;
    const int nAxes = sizeAxis ? 4 : 3;
    if (sizeAxis && !par)
      for (uint32_t i = b; i < e; ++i) sLo = std::min(sLo, prims[i].c[3]), sHi = std::max(sHi, prims[i].c[3]);
    int axisOrder[3] = {0, 1, 2};
    std::sort(axisOrder, axisOrder + 3, [&](int x, int y) { return cb.hi[x] - cb.lo[x] > cb.hi[y] - cb.lo[y]; });
    auto medianSplit = [&]() {
      const int ax = axisOrder[0];
      std::nth_element(prims.begin() + b, prims.begin() + median, prims.begin() + e,
                       [ax](const Prim& p, const Prim& q) { return p.c[ax] < q.c[ax] || (p.c[ax] == q.c[ax] && p.id < q.id); });
      return median;
    };
    // depth budget: a child at depth + 1 can still be subdivided into leaves within
    // depthCap levels iff it holds <= leafMax << (depthCap - depth - 1) primitives; SAH
    // chooses among the splits both of whose sides satisfy that (the median always does)
    const int rem = depthCap - depth - 1;
    const uint64_t maxSide = rem >= 31 ? ~0ull : (uint64_t)leafMax << (rem < 0 ? 0 : rem);

    // small ranges: exact SAH sweep over all three axes (every split position).  The orders are sorted as 12-byte (key,
    // id, position) triples, not as 44-byte primitives (the host build spends most of its time in these sorts); the
    // comparator is a total order (ids are unique), so the arrangement is the one sorting the primitives gave.
    if (n <= kSweepMax) {
      struct KI {
        float key;
        uint32_t id, pos;
      };
      float bestCostS = std::numeric_limits<float>::infinity();
      int bestAx = -1;
      uint32_t bestPos = 0;
      std::vector<float> rightArea(n);
      std::vector<KI> ki(n), bestOrder;
      for (int ax = 0; ax < nAxes; ++ax) {
        if (ax == 3 && !(sHi > sLo)) continue;
        for (uint32_t i = 0; i < n; ++i) ki[i] = KI{prims[b + i].c[ax], prims[b + i].id, i};
        std::sort(ki.begin(), ki.end(), [](const KI& p, const KI& q) { return p.key < q.key || (p.key == q.key && p.id < q.id); });
        Box acc;
        acc.reset();
        for (uint32_t i = n; i-- > 1;) {
          acc.grow(prims[b + ki[i].pos].box);
          rightArea[i] = acc.halfArea();
        }
        acc.reset();
        bool better = false;
        for (uint32_t i = 1; i < n; ++i) {
          acc.grow(prims[b + ki[i - 1].pos].box);
          float cost = acc.halfArea() * std::ceil(i / static_cast<float>(leafMax)) +
                       rightArea[i] * std::ceil((n - i) / static_cast<float>(leafMax));
          if (ax == 3) cost *= sizeBias;
          if (cost < bestCostS && i <= maxSide && n - i <= maxSide) bestCostS = cost, bestAx = ax, bestPos = i, better = true;
        }
        if (better) bestOrder = ki;
      }
      if (bestAx < 0) return medianSplit();
      // the range in the best axis' order
      std::vector<Prim> tmp(n);
      for (uint32_t i = 0; i < n; ++i) tmp[i] = prims[b + bestOrder[i].pos];
      std::copy(tmp.begin(), tmp.end(), prims.begin() + b);
      return b + bestPos;
    }

    constexpr int NBMAX = 128;
    static const int nbEnv = getenv("RT_BVH_BINS") ? atoi(getenv("RT_BVH_BINS")) : 0;
    // 64 bins per axis for ranges of 65,536 primitives and more, 16 below (RT_BVH_BINS / RT_BVH_BINS_FROM
    // override).  Measured, Mrays/s and node visits per ray: 1 M-triangle lattice 16 bins 6,620 / 38.8,
    // 64 bins 6,781 / 37.5; 8 M triangles 4,037 / 48.1 -> 4,633 / 42.5 (16 bins across an 88-cell lattice cut
    // through the cells; 32, 64 and 128 are within 2 % of each other).  On the 11.7 k-triangle mesh the two
    // binned top levels came out 5 % WORSE in node visits with 64 bins (greedy SAH is not monotone in the
    // candidate set), and the big scenes keep their gain when only the large ranges get the fine bins.
    static const int nbSplit = getenv("RT_BVH_BINS_FROM") ? atoi(getenv("RT_BVH_BINS_FROM")) : 65536;
    const int NB = (int)n < nbSplit ? 16 : nbEnv >= 2 && nbEnv <= NBMAX ? nbEnv : 64;
    float bestCost = std::numeric_limits<float>::infinity();
    int bestAxis = -1, bestBin = -1;
    // (big ranges: the bins of all axes in ONE pass, every thread its chunk, merged — min / max and counts: the same bins
    // whatever the chunking)
    struct AxisBins {
      Box bb[NBMAX];
      uint32_t cnt[NBMAX];
    };
    std::vector<AxisBins> pbins;
    if (par) {
      float axLo4[4], scale4[4];
      bool use4[4];
      for (int ax = 0; ax < 4; ++ax) {
        axLo4[ax] = ax < 3 ? cb.lo[ax] : sLo;
        const float ext = ax < 3 ? cb.hi[ax] - cb.lo[ax] : sHi - sLo;
        use4[ax] = ax < nAxes && ext > 0.f;
        scale4[ax] = use4[ax] ? NB / ext : 0.f;
      }
      const uint32_t T = std::max<uint32_t>(1u, std::min<uint32_t>(parThreads, n / 32768u));
      std::vector<AxisBins> tb((size_t)T * 4u);
      for (AxisBins& x : tb)
        for (int k = 0; k < NB; ++k) x.bb[k].reset(), x.cnt[k] = 0;
      parChunks(b, e, [&](uint32_t t, uint32_t cb0, uint32_t ce0) {
        AxisBins* mine = &tb[(size_t)t * 4u];
        for (uint32_t i = cb0; i < ce0; ++i)
          for (int ax = 0; ax < 4; ++ax) {
            if (!use4[ax]) continue;
            const int k = std::min(NB - 1, std::max(0, static_cast<int>((prims[i].c[ax] - axLo4[ax]) * scale4[ax])));
            mine[ax].bb[k].grow(prims[i].box), ++mine[ax].cnt[k];
          }
      });
      pbins.resize(4);
      for (int ax = 0; ax < 4; ++ax)
        for (int k = 0; k < NB; ++k) {
          pbins[ax].bb[k].reset(), pbins[ax].cnt[k] = 0;
          for (uint32_t t = 0; t < T; ++t) pbins[ax].bb[k].grow(tb[(size_t)t * 4u + ax].bb[k]), pbins[ax].cnt[k] += tb[(size_t)t * 4u + ax].cnt[k];
        }
    }
    for (int ax = 0; ax < nAxes; ++ax) {
      const float axLo = ax < 3 ? cb.lo[ax] : sLo;
      const float ext = ax < 3 ? cb.hi[ax] - cb.lo[ax] : sHi - sLo;
      if (!(ext > 0.f)) continue;
      Box bbLocal[NBMAX];
      uint32_t cntLocal[NBMAX] = {0};
      Box* bb = par ? pbins[ax].bb : bbLocal;
      uint32_t* cnt = par ? pbins[ax].cnt : cntLocal;
      const float scale = NB / ext;
      if (!par) {
        for (int k = 0; k < NB; ++k) bb[k].reset();
        for (uint32_t i = b; i < e; ++i) {
          int k = std::min(NB - 1, std::max(0, static_cast<int>((prims[i].c[ax] - axLo) * scale)));
          bb[k].grow(prims[i].box), ++cnt[k];
        }
      }
      float rightArea[NBMAX];
      uint32_t rightCnt[NBMAX];
      Box acc;
      acc.reset();
      uint32_t c = 0;
      for (int k = NB - 1; k > 0; --k) {
        acc.grow(bb[k]), c += cnt[k];
        rightArea[k] = acc.halfArea(), rightCnt[k] = c;
      }
      acc.reset(), c = 0;
      for (int k = 0; k < NB - 1; ++k) {
        acc.grow(bb[k]), c += cnt[k];
        if (c == 0 || rightCnt[k + 1] == 0 || c > maxSide || rightCnt[k + 1] > maxSide) continue;
        // leaves hold up to leafMax triangles: cost in units of leaf fetches
        float cost = acc.halfArea() * std::ceil(c / static_cast<float>(leafMax)) +
                     rightArea[k + 1] * std::ceil(rightCnt[k + 1] / static_cast<float>(leafMax));
        if (ax == 3) cost *= sizeBias;
        if (cost < bestCost) bestCost = cost, bestAxis = ax, bestBin = k;
      }
    }
    if (bestAxis < 0) return medianSplit();
    const float ext = bestAxis < 3 ? cb.hi[bestAxis] - cb.lo[bestAxis] : sHi - sLo;
    const float scale = NB / ext, lo = bestAxis < 3 ? cb.lo[bestAxis] : sLo;
    const int ax = bestAxis, bin = bestBin;
    auto goesLeft = [&](const Prim& p) {
      int k = std::min(NB - 1, std::max(0, static_cast<int>((p.c[ax] - lo) * scale)));
      return k <= bin;
    };
    uint32_t m;
    if (par && scratch.size() == prims.size()) {
      // stable partition through the second buffer: count per chunk, place, copy back — three passes shared by the threads
      uint32_t nl[65] = {0};
      parChunks(b, e, [&](uint32_t t, uint32_t cb0, uint32_t ce0) {
        uint32_t c = 0;
        for (uint32_t i = cb0; i < ce0; ++i) c += goesLeft(prims[i]) ? 1u : 0u;
        nl[t + 1] = c;
      });
      uint32_t chunkB[65] = {0};
      {
        const uint32_t T = std::max<uint32_t>(1u, std::min<uint32_t>(parThreads, n / 32768u));
        for (uint32_t t = 0; t <= T; ++t) chunkB[t] = T <= 1u ? (t ? e : b) : b + (uint32_t)((uint64_t)n * t / T);
        for (uint32_t t = 0; t < 64; ++t) nl[t + 1] += nl[t];
        m = b + nl[T];
      }
      parChunks(b, e, [&](uint32_t t, uint32_t cb0, uint32_t ce0) {
        uint32_t l = b + nl[t], r = m + (cb0 - b) - nl[t];  // (rights before this chunk: elements before it minus lefts before it)
        for (uint32_t i = cb0; i < ce0; ++i) {
          if (goesLeft(prims[i])) scratch[l++] = prims[i];
          else scratch[r++] = prims[i];
        }
      });
      parChunks(b, e, [&](uint32_t, uint32_t cb0, uint32_t ce0) { std::copy(scratch.begin() + cb0, scratch.begin() + ce0, prims.begin() + cb0); });
      (void)chunkB;
    } else {
      auto mid = std::partition(prims.begin() + b, prims.begin() + e, goesLeft);
      m = static_cast<uint32_t>(mid - prims.begin());
    }
    if (m == b || m == e) return medianSplit();
    // keep the tree shallow: refuse extremely lopsided SAH splits on big ranges
    const uint32_t small = std::min(m - b, e - m);
    if (n > 64 && small * 64 < n && depth > kMaxDepth / 2) return medianSplit();
    return m;
  }

  // One builder thread's output: nodes in pre-order with LOCAL indices.
  struct Sub {
    std::vector<Node> nodes;
    uint32_t maxDepth = 0;
  };

  static void setNode(Node& nd, const Box& b0, const Box& b1, int32_t c0, int32_t c1) {
    for (int a = 0; a < 3; ++a) {
      nd.lo0[a] = b0.lo[a], nd.hi0[a] = b0.hi[a];
      nd.lo1[a] = b1.lo[a], nd.hi1[a] = b1.hi[a];
    }
    nd.child[0] = c0, nd.child[1] = c1;
    nd.pad[0] = nd.pad[1] = 0;
  }

  // Builds the subtree for [b,e) into `sub`; returns a child reference (local node
  // index or leaf code) and its (padded) box.  Ranges above `grain` primitives build
  // their right half on a new thread (fork-join): a thread only touches its own
  // prims range and its own Sub, and the right block is appended behind the left one,
  // so the node numbering is the sequential pre-order whatever the thread count.
  int32_t recurse(uint32_t b, uint32_t e, int depth, Box& boxOut, Sub& sub) {
    boxOut = boundsOf(b, e);
    for (int a = 0; a < 3; ++a) boxOut.lo[a] -= out.pad, boxOut.hi[a] += out.pad;
    sub.maxDepth = std::max<uint32_t>(sub.maxDepth, depth);
    if (e - b <= leafMax) {
      // leaf: records in ascending global id so equal-t ties inside a leaf are
      // met lowest id first (not required for correctness, just tidy)
      std::sort(prims.begin() + b, prims.begin() + e, [](const Prim& p, const Prim& q) { return p.id < q.id; });
      return encodeLeaf(b, e - b);
    }
    if (cutoff && e - b <= cutoff) {  // a part: its subtree is built elsewhere (bvh_gpu.hip k_subtree)
      std::lock_guard<std::mutex> lock(partsLock);
      parts.push_back(TopBuilt::Part{b, e, (uint32_t)depth, 0u, 0u});
      return (int32_t)~(kPartFlag | (uint32_t)(parts.size() - 1));
    }
    if (depth >= kMaxDepth - 1) throw std::runtime_error("BVH depth budget exceeded");
    const uint32_t m = split(b, e, depth);
    const int32_t self = static_cast<int32_t>(sub.nodes.size());
    sub.nodes.emplace_back();
    Box b0, b1;
    int32_t c0, c1;
    if (e - b > grain) {
      Sub right;
      std::exception_ptr err;
      std::thread t([&] {
        try {
          c1 = recurse(m, e, depth + 1, b1, right);
        } catch (...) {
          err = std::current_exception();
        }
      });
      try {
        c0 = recurse(b, m, depth + 1, b0, sub);
      } catch (...) {
        t.join();
        throw;
      }
      t.join();
      if (err) std::rethrow_exception(err);
      const int32_t off = static_cast<int32_t>(sub.nodes.size());
      for (Node nd : right.nodes) {
        if (nd.child[0] >= 0) nd.child[0] += off;
        if (nd.child[1] >= 0) nd.child[1] += off;
        sub.nodes.push_back(nd);
      }
      if (c1 >= 0) c1 += off;
      sub.maxDepth = std::max(sub.maxDepth, right.maxDepth);
    } else {
      c0 = recurse(b, m, depth + 1, b0, sub);
      c1 = recurse(m, e, depth + 1, b1, sub);
    }
    setNode(sub.nodes[self], b0, b1, c0, c1);
    return self;
  }
};

TriRec makeRec(const rt_scene_desc& sc, uint32_t t, uint32_t mesh) {
  TriRec r;
  const float* p0 = sc.vertex_pos + 3 * static_cast<size_t>(sc.tri_vtx[3 * static_cast<size_t>(t) + 0]);
  const float* p1 = sc.vertex_pos + 3 * static_cast<size_t>(sc.tri_vtx[3 * static_cast<size_t>(t) + 1]);
  const float* p2 = sc.vertex_pos + 3 * static_cast<size_t>(sc.tri_vtx[3 * static_cast<size_t>(t) + 2]);
  for (int a = 0; a < 3; ++a) {
    r.p0[a] = p0[a];
    r.e1[a] = p1[a] - p0[a];  // the float subtraction Ray.cpp:11 does per test
    r.e2[a] = p2[a] - p0[a];
  }
  r.id = t, r.mesh = mesh, r.pad = 0;
  return r;
}

// ---------------------------------------------------------------- tree rotations
// Post-pass over the finished tree: at every inner node N with children (L, R), L inner
// with children (L0, L1), swapping R with L0 or L1 (and the mirror cases) is applied when
// it shrinks the surface area of L's box — the SAH cost of a tree with fixed leaves is
// the sum of its inner boxes' areas.  Children are visited before their parent; a swap
// that would push a subtree below the depth cap is not considered.  Node indices stay,
// only child links and child boxes move; the nodes are renumbered in pre-order afterwards.
struct Rotator {
  std::vector<Node>& nodes;
  int depthCap;
  std::vector<uint8_t> height;  // max leaf depth below a node, relative (a node over two leaves: 1)
  uint64_t swaps = 0;

  static Box childBox(const Node& n, int i) {
    Box b;
    for (int a = 0; a < 3; ++a) b.lo[a] = i ? n.lo1[a] : n.lo0[a], b.hi[a] = i ? n.hi1[a] : n.hi0[a];
    return b;
  }
  static void setChild(Node& n, int i, int32_t ref, const Box& b) {
    for (int a = 0; a < 3; ++a) {
      if (i) n.lo1[a] = b.lo[a], n.hi1[a] = b.hi[a];
      else n.lo0[a] = b.lo[a], n.hi0[a] = b.hi[a];
    }
    n.child[i] = ref;
  }
  int heightOf(int32_t ref) const { return ref < 0 ? 0 : height[ref]; }

  void visit(int32_t idx, int depth) {
    for (int i = 0; i < 2; ++i)
      if (nodes[idx].child[i] >= 0) visit(nodes[idx].child[i], depth + 1);
    Node& N = nodes[idx];
    float bestDelta = -1e-7f * (childBox(N, 0).halfArea() + childBox(N, 1).halfArea());
    int bestS = -1, bestG = -1;
    for (int s = 0; s < 2; ++s) {
      const int32_t l = N.child[s], r = N.child[1 - s];
      if (l < 0) continue;
      if (depth + 2 + heightOf(r) > depthCap) continue;  // r would sit one level deeper
      const Box rb = childBox(N, 1 - s);
      const float oldArea = childBox(N, s).halfArea();
      for (int g = 0; g < 2; ++g) {  // grandchild g goes up, its sibling stays with r
        Box nb = childBox(nodes[l], 1 - g);
        nb.grow(rb);
        const float delta = nb.halfArea() - oldArea;
        if (delta < bestDelta) bestDelta = delta, bestS = s, bestG = g;
      }
    }
    // grandchild <-> grandchild across the two sides (no depth change): L0 <-> R0 or R1
    int bestX = -1;
    if (N.child[0] >= 0 && N.child[1] >= 0) {
      const Node& L = nodes[N.child[0]];
      const Node& R = nodes[N.child[1]];
      const float oldArea = childBox(N, 0).halfArea() + childBox(N, 1).halfArea();
      for (int x = 0; x < 2; ++x) {  // L.child[0] swaps with R.child[x]
        Box lb = childBox(R, x), rb = childBox(L, 0);
        lb.grow(childBox(L, 1));
        rb.grow(childBox(R, 1 - x));
        const float delta = lb.halfArea() + rb.halfArea() - oldArea;
        if (delta < bestDelta) bestDelta = delta, bestX = x, bestS = -1;
      }
    }
    if (bestX >= 0) {
      Node& L = nodes[N.child[0]];
      Node& R = nodes[N.child[1]];
      const Box a0 = childBox(L, 0), bx = childBox(R, bestX);
      const int32_t ra = L.child[0], rb = R.child[bestX];
      setChild(L, 0, rb, bx);
      setChild(R, bestX, ra, a0);
      Box lb = childBox(L, 0), rbb = childBox(R, 0);
      lb.grow(childBox(L, 1));
      rbb.grow(childBox(R, 1));
      const int32_t li = N.child[0], ri = N.child[1];
      setChild(N, 0, li, lb);
      setChild(N, 1, ri, rbb);
      height[li] = (uint8_t)(1 + std::max(heightOf(L.child[0]), heightOf(L.child[1])));
      height[ri] = (uint8_t)(1 + std::max(heightOf(R.child[0]), heightOf(R.child[1])));
      ++swaps;
    } else if (bestS >= 0) {
      const int s = bestS, g = bestG;
      const int32_t l = N.child[s], r = N.child[1 - s];
      Node& L = nodes[l];
      const Box rb = childBox(N, 1 - s), gb = childBox(L, g);
      const int32_t gref = L.child[g];
      setChild(L, g, r, rb);
      Box lb = childBox(L, 0);
      lb.grow(childBox(L, 1));
      setChild(N, 1 - s, gref, gb);
      setChild(N, s, l, lb);
      height[l] = (uint8_t)(1 + std::max(heightOf(L.child[0]), heightOf(L.child[1])));
      ++swaps;
    }
    height[idx] = (uint8_t)(1 + std::max(heightOf(N.child[0]), heightOf(N.child[1])));
  }

  // pre-order renumbering (what the traversal's locality was tuned on)
  void relayout() {
    std::vector<Node> out;
    out.reserve(nodes.size());
    std::vector<int32_t> stack{0};
    std::vector<int32_t> newIndex(nodes.size(), -1);
    // first pass: assign pre-order indices
    std::vector<int32_t> order;
    order.reserve(nodes.size());
    while (!stack.empty()) {
      const int32_t i = stack.back();
      stack.pop_back();
      newIndex[i] = (int32_t)order.size();
      order.push_back(i);
      if (nodes[i].child[1] >= 0) stack.push_back(nodes[i].child[1]);
      if (nodes[i].child[0] >= 0) stack.push_back(nodes[i].child[0]);
    }
    for (int32_t i : order) {
      Node n = nodes[i];
      for (int c = 0; c < 2; ++c)
        if (n.child[c] >= 0) n.child[c] = newIndex[n.child[c]];
      out.push_back(n);
    }
    nodes.swap(out);
  }
};


// ---------------------------------------------------------------- subtree moves (the measured-cost tuner's proposals)
// After Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume hierarchies", 2013, restated:
// a subtree L is taken out of the tree (its parent goes with it, its sibling moves up) and can be put back at any
// position X as "a new node over X and L"; a branch-and-bound search from the root lists the positions that add the
// least surface area (the induced growth only increases on the way down).  The depth cap is enforced (a position
// that would push a leaf below it is not a candidate) and the leaves never change.  Round 3 ran whole passes of
// this by area alone (-6.2 / -2.3 / -0.4 % inner surface, +1.0 / +1.6 / +0.2 % node visits of the renderer's rays:
// profiles/r03_reinsertion_ab.txt — the surface-area estimate no longer predicts visits at that level); the passes
// are gone, tuneMeasured below uses the moves with a MEASURED cost.
struct Reinserter {
  struct Item {
    Box box;
    int32_t parent, child[2];  // child: item index; -1 on a leaf
    int32_t leafRef;           // the leaf's code (negative); 0 on an inner node
    uint8_t height;            // max leaf depth below (a leaf: 0)
  };
  std::vector<Item> it;
  int depthCap;
  uint32_t nInner = 0;

  void load(const std::vector<Node>& nodes) {
    nInner = (uint32_t)nodes.size();
    it.assign(nInner, Item{});
    it.reserve(2 * (size_t)nInner + 1);
    for (uint32_t i = 0; i < nInner; ++i) it[i].leafRef = 0, it[i].parent = -1;
    for (uint32_t i = 0; i < nInner; ++i)
      for (int k = 0; k < 2; ++k) {
        const int32_t ref = nodes[i].child[k];
        const Box b = Rotator::childBox(nodes[i], k);
        if (ref >= 0) {
          it[ref].box = b, it[ref].parent = (int32_t)i, it[i].child[k] = ref;
        } else {
          Item l{};
          l.box = b, l.parent = (int32_t)i, l.child[0] = l.child[1] = -1, l.leafRef = ref, l.height = 0;
          it[i].child[k] = (int32_t)it.size();
          it.push_back(l);
        }
      }
    it[0].box = it[it[0].child[0]].box;
    it[0].box.grow(it[it[0].child[1]].box);
    // heights, children before parents (explicit post-order)
    std::vector<std::pair<int32_t, int>> st{{0, 0}};
    while (!st.empty()) {
      auto& [i, phase] = st.back();
      if (it[i].child[0] < 0) {
        st.pop_back();
        continue;
      }
      if (phase < 2) {
        const int32_t c = it[i].child[phase++];
        st.push_back({c, 0});
      } else {
        it[i].height = (uint8_t)(1 + std::max(it[it[i].child[0]].height, it[it[i].child[1]].height));
        st.pop_back();
      }
    }
  }

  void store(std::vector<Node>& nodes) const {
    for (uint32_t i = 0; i < nInner; ++i)
      for (int k = 0; k < 2; ++k) {
        const Item& c = it[it[i].child[k]];
        Rotator::setChild(nodes[i], k, c.child[0] < 0 ? c.leafRef : it[i].child[k], c.box);
      }
  }

  void refit(int32_t g) {
    for (; g >= 0; g = it[g].parent) {
      Item& G = it[g];
      Box b = it[G.child[0]].box;
      b.grow(it[G.child[1]].box);
      const uint8_t h = (uint8_t)(1 + std::max(it[G.child[0]].height, it[G.child[1]].height));
      bool same = h == G.height;
      for (int a = 0; a < 3 && same; ++a) same = b.lo[a] == G.box.lo[a] && b.hi[a] == G.box.hi[a];
      if (same) break;
      G.box = b, G.height = h;
    }
  }

  struct Cand {
    float bound;  // induced + area(L): no position below this node can cost less
    float induced;
    int32_t item;
    int depth;
    bool operator<(const Cand& o) const { return bound > o.bound || (bound == o.bound && item > o.item); }  // min-heap
  };
  std::vector<Cand> heap;

  // Takes L and its parent P out (the sibling S moves up); returns S.  L must not be a child of the root.
  int32_t takeOut(int32_t l) {
    const int32_t p = it[l].parent, g = it[p].parent;
    const int sl = it[p].child[1] == l ? 1 : 0;
    const int32_t s = it[p].child[1 - sl];
    const int sp = it[g].child[1] == p ? 1 : 0;
    it[g].child[sp] = s, it[s].parent = g;
    refit(g);
    return s;
  }
  // Puts P (L's detached parent) back as a new node over (X, L).
  void putBack(int32_t l, int32_t x) {
    const int32_t p = it[l].parent;
    const int32_t xp = it[x].parent;
    const int sx = it[xp].child[1] == x ? 1 : 0;
    it[xp].child[sx] = p, it[p].parent = xp;
    it[p].child[0] = x, it[p].child[1] = l;
    it[x].parent = p, it[l].parent = p;
    it[p].box = it[x].box;
    it[p].box.grow(it[l].box);
    it[p].height = (uint8_t)(1 + std::max(it[x].height, it[l].height));
    refit(xp);
  }
  // what "a new node over (X, L)" adds to the tree's surface area: the new node + the growth of X's ancestors
  float costAt(int32_t x, const Box& lb) const {
    Box u = it[x].box;
    u.grow(lb);
    float c = u.halfArea();
    for (int32_t a = it[x].parent; a > 0; a = it[a].parent) {
      Box w = it[a].box;
      w.grow(lb);
      c += w.halfArea() - it[a].box.halfArea();
    }
    return c;
  }
  // The K best positions for the (taken-out) subtree L by added surface area, `exclude` not being a candidate
  // (best first; fewer if fewer are admissible).
  struct Pos {
    float cost;
    int32_t item;
  };
  void searchK(int32_t l, int32_t exclude, int K, std::vector<Pos>& out) {
    const Box lb = it[l].box;
    const float la = lb.halfArea();
    const int lh = it[l].height;
    out.clear();
    auto worst = [&] { return (int)out.size() < K ? std::numeric_limits<float>::infinity() : out.back().cost; };
    heap.clear();
    heap.push_back(Cand{la, 0.f, it[0].child[0], 1});
    std::push_heap(heap.begin(), heap.end());
    heap.push_back(Cand{la, 0.f, it[0].child[1], 1});
    std::push_heap(heap.begin(), heap.end());
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end());
      const Cand c = heap.back();
      heap.pop_back();
      if (c.bound >= worst()) break;
      const Item& X = it[c.item];
      if (c.depth + 1 + lh > depthCap) continue;  // L itself would end up too deep here and anywhere below
      Box u = X.box;
      u.grow(lb);
      const float direct = u.halfArea();
      const float total = c.induced + direct;
      if (total < worst() && c.item != exclude && c.depth + 1 + (int)X.height <= depthCap) {
        if ((int)out.size() == K) out.pop_back();
        size_t at = out.size();
        out.push_back(Pos{total, c.item});
        while (at > 0 && (out[at - 1].cost > total || (out[at - 1].cost == total && out[at - 1].item > c.item))) std::swap(out[at - 1], out[at]), --at;
      }
      if (X.child[0] >= 0) {
        const float ind = c.induced + direct - X.box.halfArea();
        if (ind + la < worst()) {
          heap.push_back(Cand{ind + la, ind, X.child[0], c.depth + 1});
          std::push_heap(heap.begin(), heap.end());
          heap.push_back(Cand{ind + la, ind, X.child[1], c.depth + 1});
          std::push_heap(heap.begin(), heap.end());
        }
      }
    }
  }
};

}  // namespace

// float -> binary16 with directed rounding (toward -inf when up == false, toward
// +inf when up == true).  Inputs are finite and |x| <= 32768 by construction.
uint16_t toHalfDirected(float x, bool up) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  const uint32_t sign = u >> 31;
  const float ax = std::fabs(x);
  // away-from-zero rounding of the magnitude is needed iff the direction matches the sign
  const bool away = (up && !sign) || (!up && sign);
  uint32_t h;
  if (ax == 0.f) {
    h = 0;
  } else if (ax < 6.103515625e-05f) {  // below the smallest normal half: fixed-point with 2^-24 steps
    const float q = ax * 16777216.f;   // exact
    uint32_t m = static_cast<uint32_t>(q);
    if (away && static_cast<float>(m) < q) ++m;
    h = m;                             // m == 1024 carries into the smallest normal, as it should
  } else {
    uint32_t au = u & 0x7fffffffu;
    const uint32_t lost = au & 0x1fffu;  // 13 mantissa bits do not fit
    au >>= 13;
    if (away && lost) ++au;              // a carry walks into the exponent, as it should
    h = au - ((127u - 15u) << 10);
  }
  return static_cast<uint16_t>((sign << 15) | h);
}

float halfToFloat(uint16_t hv) {
  const uint32_t sign = (hv >> 15) & 1u, e = (hv >> 10) & 31u, m = hv & 1023u;
  float v;
  if (e == 0) v = static_cast<float>(m) * 5.9604644775390625e-08f;  // 2^-24
  else v = std::ldexp(static_cast<float>(m | 1024u), static_cast<int>(e) - 25);
  return sign ? -v : v;
}

ScenePlan planSceneExact(const rt_scene_desc& sc, uint32_t leafMax, std::vector<float>& sizeKey, uint32_t threads) {
  ScenePlan P;
  if (leafMax == 0) leafMax = 2;
  if (leafMax > 8) leafMax = 8;
  P.leafMax = leafMax;
  if (sc.n_triangles == 0 || sc.n_triangles >= (1u << 28)) throw std::runtime_error("triangle count out of range");
  if (sc.mesh_tri_begin[sc.n_meshes] != sc.n_triangles || sc.mesh_vtx_begin[sc.n_meshes] != sc.n_vertices)
    throw std::runtime_error("mesh offset tables inconsistent with counts");
  for (uint32_t m = 0; m < sc.n_meshes; ++m)
    if (sc.mesh_tri_begin[m] > sc.mesh_tri_begin[m + 1]) throw std::runtime_error("mesh_tri_begin not monotone");
  uint32_t T = threads ? threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("RT_BVH_THREADS")) T = std::max(1, atoi(e));
  T = std::max(1u, std::min({T, 64u, sc.n_triangles / 32768u + 1u}));
  sizeKey.resize(sc.n_triangles);
  std::vector<float> mx(T, 0.f);
  std::vector<int> bad(T, 0);
  auto work = [&](uint32_t th) {
    const uint32_t tb = (uint32_t)((uint64_t)sc.n_triangles * th / T), te = (uint32_t)((uint64_t)sc.n_triangles * (th + 1) / T);
    uint32_t m = 0;
    while (m + 1 < sc.n_meshes && sc.mesh_tri_begin[m + 1] <= tb) ++m;
    float localMax = 0.f;
    for (uint32_t t = tb; t < te; ++t) {
      while (t >= sc.mesh_tri_begin[m + 1]) ++m;  // (empty meshes are skipped; the tables were checked above)
      float lo[3], hi[3];
      for (int k = 0; k < 3; ++k) {
        const uint32_t v = sc.tri_vtx[3 * static_cast<size_t>(t) + k];
        if (v < sc.mesh_vtx_begin[m] || v >= sc.mesh_vtx_begin[m + 1]) {
          bad[th] = 1;
          return;
        }
        const float* q = sc.vertex_pos + 3 * static_cast<size_t>(v);
        for (int a = 0; a < 3; ++a) {
          if (!std::isfinite(q[a])) {
            bad[th] = 2;
            return;
          }
          localMax = std::max(localMax, std::fabs(q[a]));
          lo[a] = k ? std::min(lo[a], q[a]) : q[a], hi[a] = k ? std::max(hi[a], q[a]) : q[a];
        }
      }
      sizeKey[t] = -std::log2(std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-30f}));  // (Builder::loadPrims' c[3])
    }
    mx[th] = localMax;
  };
  {
    std::vector<std::thread> pool;
    for (uint32_t th = 1; th < T; ++th) pool.emplace_back(work, th);
    work(0);
    for (std::thread& x : pool) x.join();
  }
  for (uint32_t th = 0; th < T; ++th) {
    if (bad[th] == 1) throw std::runtime_error("triangle references a vertex outside its mesh");
    if (bad[th] == 2) throw std::runtime_error("non-finite vertex position");
    P.maxAbs = std::max(P.maxAbs, mx[th]);
  }
  int levels = 0;
  for (uint32_t n = sc.n_triangles; n > leafMax; n = (n + 1) / 2) ++levels;
  const char* slack = getenv("RT_BVH_SLACK");
  P.depthCap = std::min(kMaxDepth - 1, levels + (slack ? atoi(slack) : defaultDepthSlack(levels)));  // (as build() / buildTop())
  float padRef = std::max(1.f, P.maxAbs);
  for (int a = 0; a < 3; ++a)
    if (std::isfinite(sc.camera.position[a])) padRef = std::max(padRef, std::fabs(sc.camera.position[a]));
  for (uint32_t l = 0; l < sc.n_lights; ++l)
    for (int a = 0; a < 3; ++a)
      if (std::isfinite(sc.lights[l].position[a])) padRef = std::max(padRef, std::fabs(sc.lights[l].position[a]));
  P.pad = 6e-5f * padRef;
  P.originBound = 16.f * padRef;
  int e = 0;
  std::frexp(32768.f / std::max(P.maxAbs + P.pad, 1e-30f), &e);
  P.boxScale = std::ldexp(1.f, std::min(std::max(e - 1, -100), 100));
  return P;
}

// Final node numbering: the kTop nodes a ray is most likely to visit first — taken
// greedily by box surface area from the root, so the set is closed under "parent of" and
// every prefix [0, K) of it is itself such a set — then the remaining subtrees in
// pre-order.  The render kernel keeps a prefix of the array in LDS (one copy per CU,
// rt_kernels.hip k_render_persist); which K it picks depends on the LDS left over.
static void relayoutTop(std::vector<Node>& nodes, uint32_t kTop) {
  const uint32_t n = static_cast<uint32_t>(nodes.size());
  if (n <= 2) return;
  auto area = [](const float* lo, const float* hi) {
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  struct Item {
    double a;
    int32_t idx;
    bool operator<(const Item& o) const { return a < o.a || (a == o.a && idx > o.idx); }  // max-heap, low index first
  };
  std::vector<int32_t> newOf(n, -1);
  std::vector<Item> heap;
  heap.push_back({1e300, 0});
  uint32_t next = 0;
  while (!heap.empty() && next < kTop) {
    std::pop_heap(heap.begin(), heap.end());
    const Item it = heap.back();
    heap.pop_back();
    newOf[it.idx] = static_cast<int32_t>(next++);
    const Node& nd = nodes[it.idx];
    if (nd.child[0] >= 0) heap.push_back({area(nd.lo0, nd.hi0), nd.child[0]}), std::push_heap(heap.begin(), heap.end());
    if (nd.child[1] >= 0) heap.push_back({area(nd.lo1, nd.hi1), nd.child[1]}), std::push_heap(heap.begin(), heap.end());
  }
  // the subtrees hanging below the top, largest first, in pre-order.  (Measured and dropped in rounds 2-3: treelets of 4 ... 512
  // nodes grown by area, sibling pairs adjacent, breadth-first below the top — all within +-0.5 % on the 1 M-triangle scene.)
  std::sort(heap.begin(), heap.end(), [](const Item& x, const Item& y) { return y < x; });
  std::vector<int32_t> st;
  for (const Item& r : heap) {
    st.push_back(r.idx);
    while (!st.empty()) {
      const int32_t root = st.back();
      st.pop_back();
      newOf[root] = static_cast<int32_t>(next++);
      if (nodes[root].child[1] >= 0) st.push_back(nodes[root].child[1]);
      if (nodes[root].child[0] >= 0) st.push_back(nodes[root].child[0]);
    }
  }
  if (next != n) throw std::runtime_error("internal error: relayout lost nodes");
  std::vector<Node> out(n);
  for (uint32_t i = 0; i < n; ++i) {
    Node nd = nodes[i];
    for (int c = 0; c < 2; ++c)
      if (nd.child[c] >= 0) nd.child[c] = newOf[nd.child[c]];
    out[newOf[i]] = nd;
  }
  nodes.swap(out);
}

// Child 0 = the child with the SMALLER box: any-hit rays of the big-scene kernels enter it first (rt_kernels.hip
// Trav::round); closest-hit traversal orders by entry distance and does not care.
static void smallerChildFirst(std::vector<Node>& nodes) {
  for (Node& n : nodes) {
    Box b0, b1;
    for (int a = 0; a < 3; ++a) b0.lo[a] = n.lo0[a], b0.hi[a] = n.hi0[a], b1.lo[a] = n.lo1[a], b1.hi[a] = n.hi1[a];
    if (b1.halfArea() < b0.halfArea()) {
      for (int a = 0; a < 3; ++a) std::swap(n.lo0[a], n.lo1[a]), std::swap(n.hi0[a], n.hi1[a]);
      std::swap(n.child[0], n.child[1]);
    }
  }
}


// ---------------------------------------------------------------- measured-cost tuning
// The surface-area heuristic prices uniformly distributed lines; the renderer's rays start on surfaces and half
// of them aim at the lights, and beyond a few percent the estimate no longer predicts their node visits (the
// reinsertion passes above: -6 % area, +1 % visits).  But the visits can be MEASURED: every tree over the same
// leaves returns the same hits, so a probe frame traces exactly the same rays whatever the tree, and its counters
// (node records fetched, triangles tested) are a deterministic cost of the tree for these rays.  tuneMeasured
// proposes moves — a subtree to the position the area search likes best other than where it is; the two children
// of a node in the other slot order (which child an any-hit ray enters first) — applies each, asks `measure`
// for the cost of b.nodes, and keeps the move only if the cost fell.  Leaves never change; no leaf ends up deeper
// than the deepest leaf of the tree as it came in (the traversal stacks were sized for it).
TuneReport tuneMeasured(Built& b, const std::function<double()>& measure, double budgetSeconds, uint32_t maxProbes, int verbose) {
  TuneReport rep{};
  const auto t0 = std::chrono::steady_clock::now();
  auto clock = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  // "elapsed" = either budget used up: seconds, or probe frames
  auto elapsed = [&] { return (maxProbes && rep.probes >= maxProbes) ? budgetSeconds : clock(); };
  if (b.nodes.size() < 3) return rep;
  Reinserter R;
  R.depthCap = (int)b.maxDepth;
  R.load(b.nodes);
  std::vector<uint8_t> flip(b.nodes.size(), 0);
  auto publish = [&] {
    R.store(b.nodes);
    smallerChildFirst(b.nodes);
    for (size_t i = 0; i < b.nodes.size(); ++i)
      if (flip[i]) {
        Node& n = b.nodes[i];
        for (int a = 0; a < 3; ++a) std::swap(n.lo0[a], n.lo1[a]), std::swap(n.hi0[a], n.hi1[a]);
        std::swap(n.child[0], n.child[1]);
      }
  };
  publish();
  double best = measure();
  rep.cost0 = best, rep.probes = 1;
  const float slack = getenv("RT_TUNE_SLACK") ? (float)atof(getenv("RT_TUNE_SLACK")) : 1.25f;
  const int topK = getenv("RT_TUNE_K") ? std::max(1, atoi(getenv("RT_TUNE_K"))) : 1;
  std::vector<Reinserter::Pos> cand;
  double gainMoves = 0, gainFlips = 0;
  uint32_t nMoves = 0, nFlips = 0;
  for (int pass = 0; pass < 64 && elapsed() < budgetSeconds; ++pass) {
    uint32_t acceptedThisPass = 0;
    // (a) subtree moves, largest parent first
    std::vector<std::pair<float, int32_t>> order;
    for (int32_t i = 1; i < (int32_t)R.it.size(); ++i)
      if (R.it[i].parent > 0) order.push_back({R.it[R.it[i].parent].box.halfArea(), i});
    std::sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
    for (const auto& o : order) {
      if (elapsed() >= budgetSeconds) break;
      const int32_t l = o.second;
      if (R.it[l].parent <= 0) continue;
      int32_t sOld = R.takeOut(l);
      R.searchK(l, sOld, topK, cand);
      const float limit = R.costAt(sOld, R.it[l].box) * slack;
      R.putBack(l, sOld);
      for (const Reinserter::Pos& q : cand) {
        if (!(q.cost <= limit) || elapsed() >= budgetSeconds) break;
        // (positions were found with L taken out; a position that is L's parent's old place has gone with it)
        if (R.it[l].parent <= 0) break;
        const int32_t sNow = R.takeOut(l);
        if (q.item == R.it[l].parent || q.item == sNow) {
          R.putBack(l, sNow);
          continue;
        }
        R.putBack(l, q.item);
        publish();
        const double m = measure();
        ++rep.probes;
        if (m < best) {
          gainMoves += best - m, ++nMoves;
          best = m, ++rep.accepted, ++acceptedThisPass;
          break;  // L has moved: its other candidates were priced for the old tree
        }
        R.takeOut(l);
        R.putBack(l, sNow);
      }
    }
    // (b) slot order of every inner node
    for (size_t i = 0; i < flip.size() && elapsed() < budgetSeconds; ++i) {
      flip[i] ^= 1;
      publish();
      const double m = measure();
      ++rep.probes;
      if (m < best) gainFlips += best - m, ++nFlips, best = m, ++rep.accepted, ++acceptedThisPass;
      else flip[i] ^= 1;
    }
    if (verbose)
      fprintf(stderr, "tune pass %d: %u accepted, cost %.6g -> %.6g, %u probes, %.2f s (so far: %u subtree moves worth %.4g, %u slot flips worth %.4g)\n",
              pass, acceptedThisPass, rep.cost0, best, rep.probes, clock(), nMoves, gainMoves, nFlips, gainFlips);
    if (!acceptedThisPass) break;
  }
  publish();
  rep.cost1 = best, rep.seconds = clock();
  b.maxDepth = R.it[0].height;
  return rep;
}

// b.nodes (float boxes) -> b.nodes16 (the device records)
void packNodes(Built& out) {
  out.nodes16.resize(out.nodes.size());
  for (size_t i = 0; i < out.nodes.size(); ++i) {
    const Node& n = out.nodes[i];
    Node16& q = out.nodes16[i];
    for (int a = 0; a < 3; ++a) {
      q.box0[2 * a] = toHalfDirected(n.lo0[a] * out.boxScale, false), q.box0[2 * a + 1] = toHalfDirected(n.hi0[a] * out.boxScale, true);
      q.box1[2 * a] = toHalfDirected(n.lo1[a] * out.boxScale, false), q.box1[2 * a + 1] = toHalfDirected(n.hi1[a] * out.boxScale, true);
      if (halfToFloat(q.box0[2 * a]) > n.lo0[a] * out.boxScale || halfToFloat(q.box0[2 * a + 1]) < n.hi0[a] * out.boxScale ||
          halfToFloat(q.box1[2 * a]) > n.lo1[a] * out.boxScale || halfToFloat(q.box1[2 * a + 1]) < n.hi1[a] * out.boxScale)
        throw std::runtime_error("internal error: packed box does not contain the float box");
    }
    // inner refs as BYTE offsets of the 32-B record (the traversal adds them to the base as they are)
    // — and so do leaves: ~(byte offset of the first 48-B record | count - 1) (the offset is a multiple of 16)
    for (int c = 0; c < 2; ++c) {
      if (n.child[c] >= 0) {
        q.child[c] = n.child[c] * 32;
      } else {
        const uint32_t code = ~(uint32_t)n.child[c];
        q.child[c] = (int32_t)~((code >> 3) * 48u | (code & 7u));
      }
    }
  }
}

// RT_BVH_VERBOSE: what the 8-bit frames cost in box surface, by tree level
static void reportQ8(const Built& b) {
  std::vector<double> aF(64, 0.0), aQ(64, 0.0);
  std::vector<uint32_t> cnt(64, 0);
  struct E { int32_t node; uint32_t off; int lvl; };
  std::vector<E> st{{0, kQ8RootOffset, 0}};
  while (!st.empty()) {
    const E e = st.back();
    st.pop_back();
    float lo[2][3], hi[2][3];
    int32_t ch[2];
    decodeQ8(b, e.off, lo, hi, ch);
    const Node& nd = b.nodes[e.node];
    for (int c = 0; c < 2; ++c) {
      const float* fl = c ? nd.lo1 : nd.lo0;
      const float* fh = c ? nd.hi1 : nd.hi0;
      const double x = (double)fh[0] - fl[0], y = (double)fh[1] - fl[1], z = (double)fh[2] - fl[2];
      const double X = (double)hi[c][0] - lo[c][0], Y = (double)hi[c][1] - lo[c][1], Z = (double)hi[c][2] - lo[c][2];
      aF[e.lvl] += x * y + y * z + z * x, aQ[e.lvl] += X * Y + Y * Z + Z * X, cnt[e.lvl]++;
      if (nd.child[c] >= 0) st.push_back({nd.child[c], (uint32_t)ch[c], e.lvl + 1});
    }
  }
  for (int l = 0; l < 64; ++l)
    if (cnt[l]) fprintf(stderr, "Q8 level %2d: %8u child boxes, area float %.6g, decoded %.6g (+%.2f %%)\n", l, cnt[l], aF[l], aQ[l], 100 * (aQ[l] / aF[l] - 1));
}

// ---------------------------------------------------------------- the one-request form (bvh_build.h Slot16)
void packQ8(Built& b, uint32_t shift) {
  if (const char* e = getenv("RT_Q8_SHIFT")) shift = (uint32_t)atoi(e);  // (experiments: slots per block = 1 << shift)
  b.q8.clear(), b.q8Shift = shift, b.q8Blocks = 0;
  if (b.leafMax > 2) throw std::runtime_error("the Q8 node format holds leaves of at most 2 triangles");
  if (shift < 4 || shift > 16) throw std::runtime_error("Q8 block shift out of range");
  const uint32_t n = static_cast<uint32_t>(b.nodes.size());
  if (n == 0) return;
  const uint32_t B = 1u << shift;
  // Blocks are TREELETS, because a block's frame is the union of the boxes its records hold and the grid step is 1 / 255
  // of it: (a) the top of a subtree too big for one block, grown from its root by box area until the block is full —
  // the frame is the root's box and the members are its upper levels; (b) whole subtrees that fit, several neighbouring
  // ones per block as long as they hang below one ancestor of at most two blocks' worth (so the union stays a compact
  // region).  Depth-first order over the whole tree (measured first) put the record of every right sibling the walk
  // returned to, levels above, among leaf-level records, and two thirds of the blocks got grids 2-16 x too coarse.
  std::vector<uint32_t> slotOf(n, ~0u), groupOf(n, ~0u), need(n, 0), tin(n, 0), tout(n, 0);
  std::vector<int32_t> parent(n, -1);
  auto itemSlots = [](int32_t ref) { return ref >= 0 ? 1u : 3u * ((~(uint32_t)ref & 7u) + 1u); };
  auto groupSlots = [&](int32_t i) { return itemSlots(b.nodes[i].child[0]) + itemSlots(b.nodes[i].child[1]); };
  {
    // pre-order numbers, parents, and the slots the groups of a whole subtree need (children before parents)
    std::vector<std::pair<int32_t, int>> st{{0, 0}};
    uint32_t clock = 0;
    while (!st.empty()) {
      auto& [i, phase] = st.back();
      if (phase == 0) tin[i] = clock++;
      if (phase < 2) {
        const int32_t c = b.nodes[i].child[phase++];
        if (c >= 0) parent[c] = i, st.push_back({c, 0});
      } else {
        uint64_t m = groupSlots(i);
        for (int k = 0; k < 2; ++k)
          if (b.nodes[i].child[k] >= 0) m += need[b.nodes[i].child[k]];
        need[i] = (uint32_t)std::min<uint64_t>(m, 0xffffffffu);
        tout[i] = clock;
        st.pop_back();
      }
    }
  }
  auto area = [](const float* lo, const float* hi) {
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  struct Item {
    double a;
    int32_t idx;
    bool operator<(const Item& o) const { return a < o.a || (a == o.a && idx > o.idx); }  // max-heap, low index first
  };
  struct Layout {
    const Built& b;
    uint32_t B;
    std::vector<uint32_t>&slotOf, &groupOf, &need, &tin, &tout;
    std::vector<int32_t>& parent;
    uint64_t pos = 0, blockEnd = 0;
    int32_t firstRoot = -1;  // the first whole subtree placed in the open block; -1: the block takes no (more) subtrees
    bool fresh = false;
    void openBlock() {
      pos = (pos + B - 1) & ~(uint64_t)(B - 1);
      blockEnd = pos + B;
      pos += 1;  // slot 0: the frame
      firstRoot = -1, fresh = true;
      if (blockEnd >= (1ull << 27)) throw std::runtime_error("the tree exceeds the 2 GiB the Q8 node format addresses");
    }
    uint32_t freeSlots() const { return (uint32_t)(blockEnd - pos); }
    static uint32_t itemSlots(int32_t ref) { return ref >= 0 ? 1u : 3u * ((~(uint32_t)ref & 7u) + 1u); }
    void placeGroup(int32_t i) {
      const Node& nd = b.nodes[i];
      const uint32_t s0 = itemSlots(nd.child[0]), s1 = itemSlots(nd.child[1]);
      const uint32_t g = (uint32_t)pos;
      pos += s0 + s1;
      groupOf[i] = g;
      if (nd.child[0] >= 0) slotOf[nd.child[0]] = g;
      if (nd.child[1] >= 0) slotOf[nd.child[1]] = g + s0;
    }
    void placeSubtree(int32_t f) {  // depth-first over sibling pairs
      std::vector<int32_t> st{f};
      while (!st.empty()) {
        const int32_t x = st.back();
        st.pop_back();
        placeGroup(x);
        if (b.nodes[x].child[1] >= 0) st.push_back(b.nodes[x].child[1]);
        if (b.nodes[x].child[0] >= 0) st.push_back(b.nodes[x].child[0]);
      }
    }
    bool compatible(int32_t f) const {
      if (fresh) return true;
      if (firstRoot < 0) return false;
      int32_t a = f;
      while (!(tin[a] <= tin[firstRoot] && tin[firstRoot] < tout[a])) a = parent[a];
      return need[a] <= 2u * (B - 1u);
    }
    // the upper levels of subtree f into the open block, by box area; what does not fit is returned (pre-order)
    std::vector<int32_t> growTop(int32_t f, const std::function<double(const float*, const float*)>& area) {
      std::vector<Item> heap{{1e300, f}};
      std::vector<int32_t> frontier;
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        const Item it = heap.back();
        heap.pop_back();
        const Node& nd = b.nodes[it.idx];
        if (itemSlots(nd.child[0]) + itemSlots(nd.child[1]) > freeSlots()) {
          frontier.push_back(it.idx);
          continue;
        }
        placeGroup(it.idx);
        if (nd.child[0] >= 0) heap.push_back({area(nd.lo0, nd.hi0), nd.child[0]}), std::push_heap(heap.begin(), heap.end());
        if (nd.child[1] >= 0) heap.push_back({area(nd.lo1, nd.hi1), nd.child[1]}), std::push_heap(heap.begin(), heap.end());
      }
      fresh = false, firstRoot = -1;
      std::sort(frontier.begin(), frontier.end(), [&](int32_t x, int32_t y) { return tin[x] < tin[y]; });
      return frontier;
    }
    void layoutList(const std::vector<int32_t>& roots, const std::function<double(const float*, const float*)>& area) {
      for (const int32_t f : roots) {
        if (need[f] <= freeSlots() && compatible(f)) {
          if (fresh) firstRoot = f, fresh = false;
          placeSubtree(f);
        } else if (need[f] <= B - 1u) {
          openBlock();
          firstRoot = f, fresh = false;
          placeSubtree(f);
        } else {
          openBlock();
          const std::vector<int32_t> frontier = growTop(f, area);
          layoutList(frontier, area);
        }
      }
    }
  } L{b, B, slotOf, groupOf, need, tin, tout, parent};
  L.openBlock();
  slotOf[0] = (uint32_t)L.pos++;
  if (slotOf[0] * 16u != kQ8RootOffset) throw std::runtime_error("internal error: Q8 root slot");
  if (need[0] <= L.freeSlots()) {
    L.placeSubtree(0);
  } else {
    const std::vector<int32_t> frontier = L.growTop(0, area);
    L.layoutList(frontier, area);
  }
  const uint64_t pos = L.pos;
  for (uint32_t i = 0; i < n; ++i)
    if (slotOf[i] == ~0u || groupOf[i] == ~0u) throw std::runtime_error("internal error: Q8 layout lost a node");
  const uint32_t nSlots = static_cast<uint32_t>(pos), nBlocks = (nSlots + B - 1) >> shift;
  b.q8.assign(nSlots, Slot16{{0, 0, 0, 0}});
  b.q8Blocks = nBlocks;
  // frames: the union of the child boxes stored in each block's records; one step for the three axes
  std::vector<Box> fb(nBlocks);
  for (Box& x : fb) x.reset();
  for (uint32_t i = 0; i < n; ++i) {
    Box& x = fb[slotOf[i] >> shift];
    x.grow(b.nodes[i].lo0), x.grow(b.nodes[i].hi0), x.grow(b.nodes[i].lo1), x.grow(b.nodes[i].hi1);
  }
  std::vector<double> step(nBlocks, 0.0);
  for (uint32_t k = 0; k < nBlocks; ++k) {
    const Box& x = fb[k];
    if (!(x.lo[0] <= x.hi[0])) continue;  // a block of triangle records only
    double ext = 0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, (double)x.hi[a] - (double)x.lo[a]);
    // the step: any float will do (a plane = origin + q * step is exact in double either way), a power of two would
    // waste up to half of the 256 levels
    float sf0 = (float)(std::max(ext, 1e-30) / 255.0);
    if ((double)sf0 * 255.0 < ext) sf0 = std::nextafter(sf0, std::numeric_limits<float>::infinity());
    double s = std::max((double)sf0, 1e-30);
    for (int a = 0; a < 3; ++a)
      while (std::ceil(((double)x.hi[a] - (double)x.lo[a]) / s) > 255.0) s = (double)std::nextafter((float)s, std::numeric_limits<float>::infinity());
    step[k] = s;
    Slot16& h = b.q8[(size_t)k << shift];
    const float sf = (float)s;
    std::memcpy(&h.w[0], &x.lo[0], 4), std::memcpy(&h.w[1], &x.lo[1], 4), std::memcpy(&h.w[2], &x.lo[2], 4), std::memcpy(&h.w[3], &sf, 4);
  }
  for (uint32_t i = 0; i < n; ++i) {
    const Node& nd = b.nodes[i];
    const uint32_t k = slotOf[i] >> shift;
    const double s = step[k];
    const float* o = fb[k].lo;
    uint8_t q[12];
    for (int c = 0; c < 2; ++c)
      for (int a = 0; a < 3; ++a) {
        const double lo = c ? nd.lo1[a] : nd.lo0[a], hi = c ? nd.hi1[a] : nd.hi0[a];
        double ql = std::floor((lo - (double)o[a]) / s), qh = std::ceil((hi - (double)o[a]) / s);
        ql = std::max(ql, 0.0);
        while (ql > 0.0 && (double)o[a] + ql * s > lo) ql -= 1.0;
        while ((double)o[a] + qh * s < hi) qh += 1.0;
        if (!((double)o[a] + ql * s <= lo) || qh > 255.0 || ql > qh) throw std::runtime_error("internal error: Q8 plane outside its frame");
        q[6 * c + 2 * a] = (uint8_t)ql, q[6 * c + 2 * a + 1] = (uint8_t)qh;
      }
    Slot16& r = b.q8[slotOf[i]];
    std::memcpy(r.w, q, 12);
    uint32_t flags = 0;
    for (int c = 0; c < 2; ++c)
      if (nd.child[c] < 0) flags |= (1u << c) | ((((~(uint32_t)nd.child[c]) & 7u) ? 4u : 0u) << c);
    r.w[3] = (groupOf[i] << 4) | flags;
    uint32_t at = groupOf[i];
    for (int c = 0; c < 2; ++c) {
      if (nd.child[c] >= 0) {
        if (slotOf[nd.child[c]] != at) throw std::runtime_error("internal error: Q8 sibling items not adjacent");
        at += 1;
      } else {
        const uint32_t code = ~(uint32_t)nd.child[c], first = code >> 3, cnt = (code & 7u) + 1u;
        for (uint32_t t = 0; t < cnt; ++t) std::memcpy(&b.q8[at + 3u * t], &b.tris[first + t], sizeof(TriRec));
        at += 3u * cnt;
      }
    }
  }
  if (getenv("RT_BVH_VERBOSE")) {
    reportQ8(b);
    std::vector<uint32_t> hist(64, 0);
    for (uint32_t k = 0; k < nBlocks; ++k)
      if (step[k] > 0) {
        int e = 0;
        std::frexp(step[k], &e);
        hist[std::min(63, std::max(0, e + 40))]++;
      }
    for (int e = 0; e < 64; ++e)
      if (hist[e]) fprintf(stderr, "Q8 blocks with step 2^%d: %u\n", e - 41, hist[e]);
  }
}

void decodeQ8(const Built& b, uint32_t byteOffset, float lo[2][3], float hi[2][3], int32_t child[2]) {
  const uint32_t slot = byteOffset >> 4;
  if ((byteOffset & 15u) || slot >= b.q8.size() || (slot & ((1u << b.q8Shift) - 1u)) == 0) throw std::runtime_error("Q8 record offset out of range");
  const Slot16& h = b.q8[(size_t)(slot >> b.q8Shift) << b.q8Shift];
  const Slot16& r = b.q8[slot];
  float o[3], s;
  std::memcpy(o, h.w, 12), std::memcpy(&s, &h.w[3], 4);
  uint8_t q[12];
  std::memcpy(q, r.w, 12);
  for (int c = 0; c < 2; ++c)
    for (int a = 0; a < 3; ++a) {
      // (exact in double: a float plus at most 255 steps of a power of two)
      const double dl = (double)o[a] + (double)q[6 * c + 2 * a] * (double)s, dh = (double)o[a] + (double)q[6 * c + 2 * a + 1] * (double)s;
      float fl = (float)dl, fh = (float)dh;
      if ((double)fl < dl) fl = std::nextafter(fl, std::numeric_limits<float>::infinity());   // report the box no LARGER than it is:
      if ((double)fh > dh) fh = std::nextafter(fh, -std::numeric_limits<float>::infinity());  // the check must hold for the exact planes
      lo[c][a] = fl, hi[c][a] = fh;
    }
  const uint32_t ref = r.w[3], base = ref & ~15u;
  const bool leaf0 = ref & 1u, leaf1 = ref & 2u;
  const uint32_t cnt0 = (ref >> 2) & 1u, cnt1 = (ref >> 3) & 1u;
  const uint32_t off1 = base + (leaf0 ? 48u * (cnt0 + 1u) : 16u);
  child[0] = leaf0 ? (int32_t)~(base | cnt0) : (int32_t)base;
  child[1] = leaf1 ? (int32_t)~(off1 | cnt1) : (int32_t)off1;
}

void relayoutAndPack(Built& b) {
  relayoutTop(b.nodes, kTopNodes);
  packNodes(b);
}

void buildTop(const rt_scene_desc& sc, uint32_t leafMax, uint32_t cutoff, TopBuilt& out, uint32_t threads) {
  if (leafMax == 0) leafMax = 2;
  if (leafMax > 8) leafMax = 8;
  if (cutoff <= leafMax) throw std::runtime_error("buildTop: the cutoff must exceed the leaf size");
  if (sc.n_triangles == 0 || sc.n_triangles >= (1u << 28)) throw std::runtime_error("triangle count out of range");
  if (sc.mesh_tri_begin[sc.n_meshes] != sc.n_triangles || sc.mesh_vtx_begin[sc.n_meshes] != sc.n_vertices)
    throw std::runtime_error("mesh offset tables inconsistent with counts");
  const auto tTop0 = std::chrono::steady_clock::now();
  auto msSince = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tTop0).count(); };
  Built scratch;  // (the Builder writes triangle records and the plan's numbers here; only the numbers are kept)
  scratch.leafMax = leafMax;
  Builder B(sc, scratch, leafMax);
  B.cutoff = cutoff;
  const char* slack = getenv("RT_BVH_SLACK");
  B.depthCap = std::min(kMaxDepth - 1, B.levelsFor(sc.n_triangles) + (slack ? atoi(slack) : defaultDepthSlack(B.levelsFor(sc.n_triangles))));
  uint32_t nthreads = threads ? threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("RT_BVH_THREADS")) nthreads = std::max(1, atoi(e));
  B.parThreads = std::min(nthreads, 64u);
  const float maxAbs = B.loadPrims(nullptr);
  if (B.parThreads > 1u && sc.n_triangles >= Builder::kParSplit) B.scratch.resize(sc.n_triangles);
  float padRef = std::max(1.f, maxAbs);
  for (int a = 0; a < 3; ++a)
    if (std::isfinite(sc.camera.position[a])) padRef = std::max(padRef, std::fabs(sc.camera.position[a]));
  for (uint32_t l = 0; l < sc.n_lights; ++l)
    for (int a = 0; a < 3; ++a)
      if (std::isfinite(sc.lights[l].position[a])) padRef = std::max(padRef, std::fabs(sc.lights[l].position[a]));
  scratch.pad = 6e-5f * padRef;  // (recurse pads the boxes with it)
  B.grain = nthreads > 1 ? std::max<uint32_t>(8192u, sc.n_triangles / (4u * nthreads)) : ~0u;
  if (sc.n_triangles <= cutoff) throw std::runtime_error("buildTop: the scene is a single part");
  const double tPrims = msSince();
  Box root;
  Builder::Sub top;
  B.recurse(0, sc.n_triangles, 0, root, top);
  const double tSplit = msSince();
  out.nodes.swap(top.nodes);
  // (slot 0 = the left range, as recurse() made it: the rotation passes — kernels, for this tree — run on the slot order build()'s
  // own run on, and bvh_gpu.hip k_rot_pack puts the smaller box into slot 0 afterwards)
  relayoutTop(out.nodes, kTopNodes);
  // parts numbered by their place in the order (the threads registered them as they came), then their referrers
  std::vector<uint32_t> idx(B.parts.size());
  for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return B.parts[x].b < B.parts[y].b; });
  std::vector<uint32_t> newOf(idx.size());
  out.parts.resize(idx.size());
  for (uint32_t k = 0; k < idx.size(); ++k) newOf[idx[k]] = k, out.parts[k] = B.parts[idx[k]];
  out.maxDepth = top.maxDepth;
  for (uint32_t i = 0; i < out.nodes.size(); ++i)
    for (int c = 0; c < 2; ++c) {
      const int32_t ref = out.nodes[i].child[c];
      if (ref < 0 && ((~(uint32_t)ref) & kPartFlag)) {
        const uint32_t k = newOf[(~(uint32_t)ref) & (kPartFlag - 1u)];
        out.nodes[i].child[c] = (int32_t)~(kPartFlag | k);
        out.parts[k].parent = i, out.parts[k].slot = (uint32_t)c;
      }
    }
  out.order.resize(sc.n_triangles);
  for (uint32_t i = 0; i < sc.n_triangles; ++i) out.order[i] = B.prims[i].id;
  out.leafMax = leafMax, out.depthCap = B.depthCap, out.pad = scratch.pad, out.originBound = 16.f * padRef;
  int e = 0;
  std::frexp(32768.f / std::max(maxAbs + out.pad, 1e-30f), &e);
  out.boxScale = std::ldexp(1.f, std::min(std::max(e - 1, -100), 100));
  if (getenv("RT_BVH_VERBOSE"))
    fprintf(stderr, "buildTop: primitives %.1f ms, splits %.1f ms, numbering %.1f ms (%u threads)\n", tPrims, tSplit - tPrims, msSince() - tSplit, nthreads);
}

void build(const rt_scene_desc& sc, uint32_t leafMax, Built& out, uint32_t threads) {
  if (leafMax == 0) leafMax = 2;  // measured on C2: 2 -> 7.35, 3 -> 7.26, 4 -> 6.63, 8 -> 4.9 Grays/s
  if (leafMax > 8) leafMax = 8;
  if (sc.n_triangles == 0 || sc.n_triangles >= (1u << 28)) throw std::runtime_error("triangle count out of range");
  if (sc.mesh_tri_begin[sc.n_meshes] != sc.n_triangles || sc.mesh_vtx_begin[sc.n_meshes] != sc.n_vertices)
    throw std::runtime_error("mesh offset tables inconsistent with counts");
  out.nodes.clear(), out.tris.clear(), out.trisRef.clear();
  out.leafMax = leafMax, out.maxDepth = 0;

  Builder B(sc, out, leafMax);
  // Depth cap = balanced depth + 3.  Every level costs each wave 256 B of LDS stack, which
  // is what limits occupancy on big scenes; 3 spare levels keep SAH within 1 % of the
  // unconstrained tree (stress scene: depth 25 -> 22, cost 2203 -> 2221; 2 spare: 2562).
  const char* slack = getenv("RT_BVH_SLACK");
  // (bvh_build.h defaultDepthSlack: 3; deep trees 2 plus the levels that do not cost a wave)
  const int defSlack = defaultDepthSlack(B.levelsFor(sc.n_triangles));
  B.depthCap = std::min(kMaxDepth - 1, B.levelsFor(sc.n_triangles) + (slack ? atoi(slack) : defSlack));
  // threads: 0 = one per hardware thread (at most 16).  The top 6 levels fork; a single big split shares its passes.
  uint32_t nthreads = threads ? threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("RT_BVH_THREADS")) nthreads = std::max(1, atoi(e));
  B.parThreads = std::min(nthreads, 64u);
  out.trisRef.resize(sc.n_triangles);
  const float maxAbs = B.loadPrims(out.trisRef.data());
  if (B.parThreads > 1u && sc.n_triangles >= Builder::kParSplit) B.scratch.resize(sc.n_triangles);
  // The padding must dominate the float triangle test's own error, which grows with the
  // distance of the ray ORIGIN from the geometry (tvec = o - p0 rounds to ulp(|o|)): the
  // origins the integrator uses are the camera, the lights (photon emission) and surface
  // points, so they all enter the reference magnitude.  rt_trace rays from farther away
  // than originBound run the exhaustive loop instead (k_trace).
  float padRef = std::max(1.f, maxAbs);
  for (int a = 0; a < 3; ++a)
    if (std::isfinite(sc.camera.position[a])) padRef = std::max(padRef, std::fabs(sc.camera.position[a]));
  for (uint32_t l = 0; l < sc.n_lights; ++l)
    for (int a = 0; a < 3; ++a)
      if (std::isfinite(sc.lights[l].position[a])) padRef = std::max(padRef, std::fabs(sc.lights[l].position[a]));
  out.pad = 6e-5f * padRef;
  out.originBound = 16.f * padRef;

  B.grain = nthreads > 1 ? std::max<uint32_t>(8192u, sc.n_triangles / (4u * nthreads)) : ~0u;

  Box root;
  Builder::Sub top;
  if (sc.n_triangles <= leafMax) {
    // A root node is always present (child boxes live in the parent).  The slab
    // test cannot express an "empty" box, so tiny scenes get two real leaves:
    // the two halves, or the single triangle twice (a duplicate test is a no-op
    // under the strict tie-break).
    top.nodes.emplace_back();
    const uint32_t n = sc.n_triangles, half = n >= 2 ? n / 2 : 1;
    Box b0, b1;
    const int32_t c0 = B.recurse(0, half, 1, b0, top);
    const int32_t c1 = n >= 2 ? B.recurse(half, n, 1, b1, top) : B.recurse(0, 1, 1, b1, top);
    Builder::setNode(top.nodes[0], b0, b1, c0, c1);
  } else {
    B.recurse(0, sc.n_triangles, 0, root, top);
  }
  out.nodes.swap(top.nodes);
  out.maxDepth = top.maxDepth;
  {
    // rotation passes (RT_BVH_ROT overrides; 0 = off).  Measured with 4: nodes/ray 9.06 -> 8.71
    // on C2 (+1.9 %), 38.1 -> 37.2 on C5 (+1.1 %); SAH cost -3 % / -0.3 % / -2.4 % (1.2 k /
    // 11.7 k / 1 M triangles) after 8.
    const char* e = getenv("RT_BVH_ROT");
    const int passes = e ? atoi(e) : (sc.n_triangles > 200000u ? 3 : 8);  // (0.1 s per pass and million triangles)
    if (passes > 0 && sc.n_triangles > leafMax) {
      Rotator R{out.nodes, B.depthCap, std::vector<uint8_t>(out.nodes.size(), 0)};
      for (int p = 0; p < passes; ++p) {
        const uint64_t before = R.swaps;
        R.visit(0, 0);
        if (R.swaps == before) break;
      }
      R.relayout();
      out.maxDepth = R.height[0];  // (index 0 is the root before and after the relayout)
    }
  }
  smallerChildFirst(out.nodes);
  relayoutTop(out.nodes, kTopNodes);
  if (const char* e = getenv("RT_BVH_TUNE_AREA")) {
    // The tuner (tuneMeasured) driven by a cost the host can compute — the summed surface area of the child boxes —
    // instead of a probe frame's counters: no use for rendering (that is what the reinsertion passes do, better), but
    // it runs the tuner's whole machinery (proposals, undo, slot flips, depth bound) without a GPU: tests/test_bvh_wide_host.py.
    const uint32_t probes = (uint32_t)std::max(0, atoi(e));
    if (probes) {
      auto area = [&out]() {
        double a = 0;
        for (const Node& n : out.nodes) a += Rotator::childBox(n, 0).halfArea() + Rotator::childBox(n, 1).halfArea();
        return a;
      };
      tuneMeasured(out, area, 1e9, probes, getenv("RT_BVH_VERBOSE") != nullptr);
      relayoutTop(out.nodes, kTopNodes);
    }
  }
  out.tris.resize(sc.n_triangles);
  for (uint32_t i = 0; i < sc.n_triangles; ++i) out.tris[i] = out.trisRef[B.prims[i].id];

  // packed device nodes
  int e = 0;
  std::frexp(32768.f / std::max(maxAbs + out.pad, 1e-30f), &e);  // value = m * 2^e, m in [0.5,1)
  out.boxScale = std::ldexp(1.f, std::min(std::max(e - 1, -100), 100));
  out.depthCap = B.depthCap;
  packNodes(out);
  if (getenv("RT_BVH_VERBOSE")) {
    // what the 16-bit plane encodings cost in box surface (the traversal visits boxes roughly in proportion to it):
    // binary16 of coordinate x boxScale (what the kernels read) against a 16-bit fixed-point grid over the scene
    double aF = 0, aH = 0, aX = 0;
    int ex = 0;
    std::frexp((maxAbs + out.pad) * 2.f / 65534.f, &ex);
    const float step = std::ldexp(1.f, ex);  // power of two >= range / 65534
    for (size_t i = 0; i < out.nodes.size(); ++i)
      for (int c = 0; c < 2; ++c) {
        const Box b = Rotator::childBox(out.nodes[i], c);
        const uint16_t* q = c ? out.nodes16[i].box1 : out.nodes16[i].box0;
        Box h, x;
        for (int a = 0; a < 3; ++a) {
          h.lo[a] = halfToFloat(q[2 * a]) / out.boxScale, h.hi[a] = halfToFloat(q[2 * a + 1]) / out.boxScale;
          x.lo[a] = std::floor(b.lo[a] / step) * step, x.hi[a] = std::ceil(b.hi[a] / step) * step;
        }
        aF += b.halfArea(), aH += h.halfArea(), aX += x.halfArea();
      }
    fprintf(stderr, "box surface: float %.6g, binary16 planes %.6g (+%.2f %%), 16-bit fixed point (step %g) %.6g (+%.2f %%)\n", aF, aH,
            100 * (aH / aF - 1), (double)step, aX, 100 * (aX / aF - 1));
  }
}

}  // namespace rtbvh
