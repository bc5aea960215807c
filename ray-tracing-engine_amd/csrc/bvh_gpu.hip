// bvh_gpu.hip — the scene BVH built on the device.
//
// The reference has no usable BVH (source/BVH.h, AABB.cpp: dead code), so
// like the host builder (bvh_build.cpp) this is our own design; it emits the SAME arrays
// the traversal kernels read — 32-B binary16-packed nodes holding both child boxes, 48-B
// triangle records in leaf order, leaves of <= leafMax triangles, depth <= the plan's cap —
// so every exactness test runs unchanged on a device-built tree (RT_BVH_GPU=1).
//
//   1. per-triangle boxes + centroids, centroid bounds (one pass, ordered-int atomics);
//   2. sort key = 2-bit SIZE CLASS (triangles whose box spans > 1/4, 1/16, 1/64 of the scene
//      extent come first: a wall quad must not sit in the middle of a mesh's Morton range,
//      where every range box containing it would be the whole room) + 60-bit Morton code of
//      the centroid; radix sort (rocPRIM) of (key, triangle); equal keys keep ascending
//      triangle order (stable sort): deterministic;
//   3. a min/max segment tree over the sorted triangle boxes: the box of ANY contiguous
//      range of the Morton order in ~2 log n steps;
//   4. top-down, one launch set per tree level, one WAVE per node: the 64 lanes evaluate
//      the SAH cost (area x ceil(n / leafMax), the host builder's) of up to 64 split
//      positions of the node's Morton range — every position for ranges of <= 65
//      triangles, and always the positions where the size class changes — restricted to
//      splits whose sides still fit the remaining depth budget, and a wave-min picks the
//      cheapest (lowest position on ties);
//   5. children are numbered by an exclusive scan over the level (breadth-first node
//      order: the top of the tree is a prefix of the array, as the LDS-resident top wants),
//      nodes are written packed with the plan's padding and plane scale.
//   6. (round 3) ranges of <= kSubMax triangles leave the Morton order: ONE WORKGROUP per range builds
//      the whole subtree with the host builder's exact sweep — per level and axis a bitonic sort of
//      the range's segments by centroid (LDS), segmented prefix / suffix box scans, the SAH cost of
//      every split position, an atomic min per segment — and reorders the range's triangles; the
//      subtrees are then numbered behind the top (scan of their node counts) and packed.
//      Measured: a Morton-cut top over exact subtrees of <= 1,024 triangles is within
//      -2 ... +4 % of the host tree's node visits per ray; Morton cuts all the way down cost +19 ... 29 %.
// Quality: measured nodes/ray vs the host SAH tree are in DESIGN.md.  Cost: a few ms for 1 M triangles.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdio>
#include <cstring>  // (rocPRIM's headers use memset without including it)
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "bvh_build.h"
#include "rt_kernels.h"

namespace rtk {
namespace {

struct WorkItem {
  uint32_t b, e;  // Morton-order range
};
// a range small enough for the exact builder (k_subtree): its place in the tree is patched in afterwards
struct SubItem {
  uint32_t b, e;
  uint32_t depth;        // depth of the subtree's root node
  uint32_t parent;       // top node that refers to it (~0u: the subtree IS the tree), and which child
  uint32_t slot;
};
constexpr uint32_t kSubMax = 1024;  // triangles per exact subtree (one workgroup, one triangle per thread)

__device__ __forceinline__ int fkey(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float funkey(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void k_init_bounds(int* cb) {
  if (threadIdx.x < 3) cb[threadIdx.x] = 0x7fffffff;        // min
  else if (threadIdx.x < 6) cb[threadIdx.x] = (int)0x80000000;  // max
}

// triangle boxes (float4 lo, hi) in REFERENCE order, centroid bounds
__global__ void k_tri_boxes(const float* __restrict__ vpos, const uint4* __restrict__ triShade, uint32_t n,
                            float4* __restrict__ lo, float4* __restrict__ hi, int* __restrict__ cb) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  float c[3] = {0.f, 0.f, 0.f};
  const bool on = t < n;
  if (on) {
    const uint4 tv = triShade[t];
    float l[3], h[3];
    for (int a = 0; a < 3; ++a) {
      const float p0 = vpos[3 * (size_t)tv.x + a], p1 = vpos[3 * (size_t)tv.y + a], p2 = vpos[3 * (size_t)tv.z + a];
      l[a] = fminf(p0, fminf(p1, p2)), h[a] = fmaxf(p0, fmaxf(p1, p2));
      c[a] = 0.5f * l[a] + 0.5f * h[a];
    }
    lo[t] = make_float4(l[0], l[1], l[2], 0.f), hi[t] = make_float4(h[0], h[1], h[2], 0.f);
  }
  for (int a = 0; a < 3; ++a) {
    int mn = on ? fkey(c[a]) : 0x7fffffff, mx = on ? fkey(c[a]) : (int)0x80000000;
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, __shfl_xor(mn, off, 64));
      mx = max(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) atomicMin(&cb[a], mn), atomicMax(&cb[3 + a], mx);
  }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // 21 bits -> every third bit
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n, const int* __restrict__ cb,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float4 l = lo[t], h = hi[t];
  const float c[3] = {0.5f * l.x + 0.5f * h.x, 0.5f * l.y + 0.5f * h.y, 0.5f * l.z + 0.5f * h.z};
  const float d[3] = {h.x - l.x, h.y - l.y, h.z - l.z};
  uint64_t code = 0;
  float sceneExt = 0.f, triExt = 0.f;
  for (int a = 0; a < 3; ++a) {
    const float mn = funkey(cb[a]), mx = funkey(cb[3 + a]);
    const float ext = mx - mn;
    sceneExt = fmaxf(sceneExt, ext), triExt = fmaxf(triExt, d[a]);
    float u = ext > 0.f ? (c[a] - mn) / ext : 0.f;
    u = fminf(fmaxf(u, 0.f), 1.f);
    const uint32_t q = (uint32_t)fminf(u * 1048576.f, 1048575.f);  // 20 bits per axis
    code |= spread21(q) << (2 - a);
  }
  // size class: 0 = spans more than a quarter of the scene ... 3 = the small rest
  const float rel = sceneExt > 0.f ? triExt / sceneExt : 0.f;
  const uint64_t cls = rel > 0.25f ? 0u : rel > 0.0625f ? 1u : rel > 0.015625f ? 2u : 3u;
  keys[t] = (cls << 60) | code, vals[t] = t;
}

// segment tree over the sorted triangle boxes: seg[N2 + i] = box of sorted triangle i
__global__ void k_seg_leaves(const float4* __restrict__ lo, const float4* __restrict__ hi, const uint32_t* __restrict__ order,
                             uint32_t n, uint32_t N2, float4* __restrict__ segLo, float4* __restrict__ segHi) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N2) return;
  const float inf = __int_as_float(0x7f800000);
  segLo[N2 + i] = i < n ? lo[order[i]] : make_float4(inf, inf, inf, 0.f);
  segHi[N2 + i] = i < n ? hi[order[i]] : make_float4(-inf, -inf, -inf, 0.f);
}
__global__ void k_seg_level(uint32_t first, uint32_t count, float4* __restrict__ segLo, float4* __restrict__ segHi) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t j = first + i;
  const float4 a = segLo[2 * j], b = segLo[2 * j + 1], c = segHi[2 * j], d = segHi[2 * j + 1];
  segLo[j] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
  segHi[j] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0.f);
}

struct Box3 {
  float lx, ly, lz, hx, hy, hz;
};
__device__ __forceinline__ Box3 range_box(const float4* __restrict__ segLo, const float4* __restrict__ segHi, uint32_t N2,
                                          uint32_t b, uint32_t e) {
  const float inf = __int_as_float(0x7f800000);
  Box3 r{inf, inf, inf, -inf, -inf, -inf};
  auto eat = [&](uint32_t j) {
    const float4 l = segLo[j], h = segHi[j];
    r.lx = fminf(r.lx, l.x), r.ly = fminf(r.ly, l.y), r.lz = fminf(r.lz, l.z);
    r.hx = fmaxf(r.hx, h.x), r.hy = fmaxf(r.hy, h.y), r.hz = fmaxf(r.hz, h.z);
  };
  for (uint32_t l = b + N2, rr = e + N2; l < rr; l >>= 1, rr >>= 1) {
    if (l & 1u) eat(l++);
    if (rr & 1u) eat(--rr);
  }
  return r;
}
__device__ __forceinline__ float half_area(const Box3& b) {
  const float dx = b.hx - b.lx, dy = b.hy - b.ly, dz = b.hz - b.lz;
  return dx < 0.f ? 0.f : dx * dy + dy * dz + dz * dx;
}

// child ref of the packed record (rtbvh::Node16): inner = byte offset of the 32-B record; leaf =
// ~(byte offset of its first 48-B triangle record | count - 1)
__device__ inline uint32_t packed_ref(int32_t ref) {
  if (ref >= 0) return (uint32_t)ref * 32u;
  const uint32_t code = ~(uint32_t)ref;
  return ~((code >> 3) * 48u | (code & 7u));
}

// float -> binary16 bits with directed rounding (bvh_build.cpp toHalfDirected, same bits)
__device__ __forceinline__ uint32_t half_directed(float x, bool up) {
  const uint32_t u = __float_as_uint(x);
  const uint32_t sign = u >> 31;
  const float ax = fabsf(x);
  const bool away = (up && !sign) || (!up && sign);
  uint32_t h;
  if (ax == 0.f) {
    h = 0;
  } else if (ax < 6.103515625e-05f) {
    const float q = ax * 16777216.f;
    uint32_t m = (uint32_t)q;
    if (away && (float)m < q) ++m;
    h = m;
  } else {
    uint32_t au = u & 0x7fffffffu;
    const uint32_t lost = au & 0x1fffu;
    au >>= 13;
    if (away && lost) ++au;
    h = au - ((127u - 15u) << 10);
  }
  return (sign << 15) | h;
}

// One wave per node of this level: choose the split, count the inner children.
__global__ __launch_bounds__(256) void k_level_split(const WorkItem* __restrict__ items, uint32_t count, uint32_t depth, int depthCap,
                                                     uint32_t leafMax, const float4* __restrict__ segLo,
                                                     const float4* __restrict__ segHi, uint32_t N2, const uint64_t* __restrict__ keys,
                                                     uint32_t* __restrict__ splitPos, uint32_t* __restrict__ innerCnt,
                                                     uint32_t* __restrict__ subCnt, uint32_t subMax) {
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (w >= count) return;
  const uint32_t b = items[w].b, e = items[w].e, n = e - b;
  // a child at depth + 1 can hold at most leafMax << (depthCap - depth - 1) triangles
  const int rem = depthCap - (int)depth - 1;
  const uint64_t maxSide = rem >= 31 ? ~0ull : (uint64_t)leafMax << (rem < 0 ? 0 : rem);
  const uint32_t kmin = (uint64_t)n > maxSide ? (uint32_t)(e - maxSide) : b + 1u;
  const uint32_t kmax = (uint64_t)n > maxSide ? (uint32_t)(b + maxSide) : e - 1u;
  const uint32_t kminC = kmin < b + 1u ? b + 1u : kmin, kmaxC = kmax > e - 1u ? e - 1u : kmax;
  float cost = __int_as_float(0x7f800000);
  uint32_t k = b + n / 2u;
  auto sah = [&](uint32_t kk) {
    const Box3 L = range_box(segLo, segHi, N2, b, kk), R = range_box(segLo, segHi, N2, kk, e);
    const float nl = (float)((kk - b + leafMax - 1u) / leafMax), nr = (float)((e - kk + leafMax - 1u) / leafMax);
    float c = half_area(L) * nl + half_area(R) * nr;
    return c == c ? c : 3.0e38f;  // (NaN-proof: a degenerate box product)
  };
  auto lower_bound = [&](uint64_t key) {
    uint32_t lo_ = b, hi_ = e;
    while (lo_ < hi_) {
      const uint32_t mid = lo_ + (hi_ - lo_) / 2u;
      if (keys[mid] < key) lo_ = mid + 1u;
      else hi_ = mid;
    }
    return lo_;
  };
  if (kminC <= kmaxC) {
    const uint32_t span = kmaxC - kminC;  // candidates kminC .. kmaxC
    if (span < 64u) {
      // every position
      if (lane <= span) k = kminC + lane, cost = sah(k);
    } else {
      // (a) evenly spaced positions; the last three lanes take the positions where the size
      // class changes inside the range
      k = kminC + (uint32_t)(((uint64_t)lane * span) / 63u);
      if (lane >= 61u) {
        const uint32_t kb = lower_bound((uint64_t)(lane - 60u) << 60);  // first key of class 1, 2, 3
        if (kb >= kminC && kb <= kmaxC) k = kb;
      }
      cost = sah(k);
      // (b) Morton-cell boundaries: a contiguous range of the curve that holds a sliver of the
      // neighbouring cell has that cell's extent in its box, so the cuts that matter are the
      // cell boundaries — the 63 places where the 6 key bits below the range's common prefix
      // change (two octree levels; lane 32 is the classic LBVH split)
      const uint64_t kf = keys[b], kl = keys[e - 1u];
      if (kf != kl && lane >= 1u) {
        const int hb = 63 - __clzll((long long)(kf ^ kl));  // highest differing bit
        const int sh = hb >= 5 ? hb - 5 : 0;
        const uint64_t prefix = hb >= 63 ? 0ull : (kf >> (hb + 1)) << (hb + 1);
        const uint64_t sub = (uint64_t)lane << sh;
        if (hb >= 5 || lane < (1u << (hb + 1))) {
          const uint32_t kb = lower_bound(prefix | sub);
          if (kb >= kminC && kb <= kmaxC) {
            const float cb_ = sah(kb);
            if (cb_ < cost || (cb_ == cost && kb < k)) cost = cb_, k = kb;
          }
        }
      }
    }
  }
  // wave-min of (cost, position): lowest position among equal costs
  unsigned long long key = ((unsigned long long)__float_as_uint(cost) << 32) | k;  // costs are >= 0: bit order == value order
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 64);
    key = o < key ? o : key;
  }
  if (lane == 0) {
    uint32_t kk = (uint32_t)key;
    if ((uint32_t)(key >> 32) >= 0x7f800000u) kk = b + n / 2u;  // no finite candidate: the median always fits the budget
    splitPos[w] = kk;
    // children: leaves (<= leafMax), exact subtrees (<= subMax), or work items of the next level
    const uint32_t n0 = kk - b, n1 = e - kk;
    innerCnt[w] = (n0 > leafMax && n0 > subMax ? 1u : 0u) + (n1 > leafMax && n1 > subMax ? 1u : 0u);
    subCnt[w] = (n0 > leafMax && n0 <= subMax ? 1u : 0u) + (n1 > leafMax && n1 <= subMax ? 1u : 0u);
  }
}

// Write the node (packed + float form) and the next level's work items.
__global__ void k_level_emit(const WorkItem* __restrict__ items, uint32_t count, uint32_t levelBase, uint32_t nextBase,
                             const uint32_t* __restrict__ splitPos, const uint32_t* __restrict__ innerOff, uint32_t leafMax,
                             const float4* __restrict__ segLo, const float4* __restrict__ segHi, uint32_t N2, float pad,
                             float boxScale, uint4* __restrict__ nodes16, float4* __restrict__ nodesF, WorkItem* __restrict__ next,
                             const uint32_t* __restrict__ subOff, uint32_t subBase, SubItem* __restrict__ subs, uint32_t subMax,
                             uint32_t depth) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= count) return;
  const uint32_t b = items[w].b, e = items[w].e, k = splitPos[w];
  uint32_t slot = innerOff[w], sslot = subBase + subOff[w];
  int32_t child[2];
  const uint32_t cb[2] = {b, k}, ce[2] = {k, e};
  Box3 B0 = range_box(segLo, segHi, N2, b, k), B1 = range_box(segLo, segHi, N2, k, e);
  // child 0 = the smaller box (any-hit rays of the big-scene kernels enter it first: rt_kernels.hip Trav::round)
  const bool swp = half_area(B1) < half_area(B0);
  if (swp) {
    const Box3 t = B0;
    B0 = B1, B1 = t;
  }
  for (int c = 0; c < 2; ++c) {
    const int o = swp ? 1 - c : c;  // the record's slot of range c
    if (ce[c] - cb[c] > leafMax && ce[c] - cb[c] <= subMax) {
      subs[sslot] = SubItem{cb[c], ce[c], depth + 1u, levelBase + w, (uint32_t)o};
      child[o] = 0;  // (patched by k_sub_relocate once the subtree's root has its index)
      ++sslot;
    } else if (ce[c] - cb[c] > leafMax) {
      next[slot] = WorkItem{cb[c], ce[c]};
      child[o] = (int32_t)(nextBase + slot);
      ++slot;
    } else {
      child[o] = ~(int32_t)((cb[c] << 3) | (ce[c] - cb[c] - 1u));
    }
  }
  B0.lx -= pad, B0.ly -= pad, B0.lz -= pad, B0.hx += pad, B0.hy += pad, B0.hz += pad;
  B1.lx -= pad, B1.ly -= pad, B1.lz -= pad, B1.hx += pad, B1.hy += pad, B1.hz += pad;
  const uint32_t i = levelBase + w;
  // float form (rtbvh::Node: lo0 hi0 lo1 hi1 child[2] pad[2])
  nodesF[4 * (size_t)i + 0] = make_float4(B0.lx, B0.ly, B0.lz, B0.hx);
  nodesF[4 * (size_t)i + 1] = make_float4(B0.hy, B0.hz, B1.lx, B1.ly);
  nodesF[4 * (size_t)i + 2] = make_float4(B1.lz, B1.hx, B1.hy, B1.hz);
  nodesF[4 * (size_t)i + 3] = make_float4(__int_as_float(child[0]), __int_as_float(child[1]), 0.f, 0.f);
  // packed form (rtbvh::Node16: per child (lo, hi) pairs for x, y, z as binary16 of coordinate x boxScale, outward)
  const float s = boxScale;
  const uint32_t h[12] = {half_directed(B0.lx * s, false), half_directed(B0.hx * s, true), half_directed(B0.ly * s, false),
                          half_directed(B0.hy * s, true),  half_directed(B0.lz * s, false), half_directed(B0.hz * s, true),
                          half_directed(B1.lx * s, false), half_directed(B1.hx * s, true), half_directed(B1.ly * s, false),
                          half_directed(B1.hy * s, true),  half_directed(B1.lz * s, false), half_directed(B1.hz * s, true)};
  nodes16[2 * (size_t)i + 0] = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
  nodes16[2 * (size_t)i + 1] = make_uint4(h[8] | h[9] << 16, h[10] | h[11] << 16, packed_ref(child[0]), packed_ref(child[1]));
}

// 48-B triangle records (rtbvh::TriRec): p0, e1 = p1 - p0, e2 = p2 - p0 (the float subtraction
// Ray.cpp:11 performs per test), global id, mesh; in `order` (leaf order) or reference order
__global__ void k_tri_records(const float* __restrict__ vpos, const uint4* __restrict__ triShade, const uint32_t* __restrict__ order,
                              uint32_t n, float4* __restrict__ recs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = order ? order[i] : i;
  const uint4 tv = triShade[t];
  const float* p0 = vpos + 3 * (size_t)tv.x;
  const float* p1 = vpos + 3 * (size_t)tv.y;
  const float* p2 = vpos + 3 * (size_t)tv.z;
  const float e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e1z = p1[2] - p0[2];
  const float e2x = p2[0] - p0[0], e2y = p2[1] - p0[1], e2z = p2[2] - p0[2];
  recs[3 * (size_t)i + 0] = make_float4(p0[0], p0[1], p0[2], e1x);
  recs[3 * (size_t)i + 1] = make_float4(e1y, e1z, e2x, e2y);
  recs[3 * (size_t)i + 2] = make_float4(e2z, __uint_as_float(t), __uint_as_float(tv.w), 0.f);
}

// ---------------------------------------------------------------- exact subtrees (step 6)
// One workgroup of T threads builds the whole subtree over a range of n <= T triangles, one triangle
// per thread, level by level; all segments (sub-ranges that still need a split) of a level are handled
// at once.  Per level:
//   for each axis: bitonic sort of the positions by (segment, centroid, position) — segments keep their
//     index ranges, so this sorts every segment by itself —, segmented suffix / prefix scans of the boxes
//     in sorted order, the host builder's cost `area(left) * ceil(nl / leafMax) + area(right) * ceil(nr /
//     leafMax)` at every split position that still fits the depth budget, atomic min per segment of
//     (cost, axis, position) — lowest cost, then lowest axis, then lowest position: bvh_build.cpp's sweep;
//   then every triangle moves to its rank along its segment's best axis, the segments split, their child
//   boxes come from one more segmented scan, new inner nodes are numbered by a scan (breadth-first inside
//   the subtree: deterministic), the node records go to the subtree's scratch block.
// Leaves keep their triangles in ascending id order (as the host builder's).
struct SubShared {
  // laid out by subtree_lds(): see k_subtree
};

template <int T>
__global__ __launch_bounds__(T) void k_subtree(const SubItem* __restrict__ items, const uint32_t* __restrict__ scratchOff,
                                               uint32_t* __restrict__ order, const float4* __restrict__ triLo,
                                               const float4* __restrict__ triHi, uint32_t leafMax, int depthCap,
                                               float4* __restrict__ scratch, uint32_t* __restrict__ cntOut,
                                               uint32_t* __restrict__ heightOut) {
  static_assert(T <= 1024 && (T & (T - 1)) == 0, "positions are 10-bit payloads of the sort keys; the bitonic network wants a power of two");
  extern __shared__ unsigned long long sub_lds[];
  // LDS carve-up (T = 1024: 8 + 8 + 36 + 4 + 24 + 4 + 6 + 10 KB = 100 KB)
  unsigned long long* key = sub_lds;                  // [T] sort keys
  unsigned long long* best = key + T;                 // [T] per segment start: (cost bits << 32 | axis << 16 | split position)
  float* eLo = reinterpret_cast<float*>(best + T);    // [3][T] element boxes, centroids (position order)
  float* eHi = eLo + 3 * T;
  float* eCen = eHi + 3 * T;
  uint32_t* eTid = reinterpret_cast<uint32_t*>(eCen + 3 * T);  // [T] triangle (reference index)
  float* sc = reinterpret_cast<float*>(eTid + T);     // [6][T] scan buffer
  float* sufA = sc + 6 * T;                           // [T] area of the suffix box
  uint16_t* rnk = reinterpret_cast<uint16_t*>(sufA + T);  // [3][T] rank of each position along each axis
  uint16_t* segS = rnk + 3 * T;                       // [T] segment [segS, segE) of the element at this position
  uint16_t* segE = segS + T;
  uint16_t* segNode = segE + T;                       // [T] local node index of the segment (all its positions carry it)
  uint16_t* segDep = segNode + T;                     // [T] depth of that node
  uint16_t* scan16 = segDep + T;                      // [T] flag scan
  __shared__ uint32_t nodeCount, anySplit, maxDep;

  const SubItem it = items[blockIdx.x];
  const uint32_t n = it.e - it.b, i = threadIdx.x;
  float4* const out = scratch + 4 * (size_t)scratchOff[blockIdx.x];
  const float inf = __int_as_float(0x7f800000);
  if (i < n) {
    const uint32_t t = order[it.b + i];
    const float4 l = triLo[t], h = triHi[t];
    eLo[i] = l.x, eLo[T + i] = l.y, eLo[2 * T + i] = l.z;
    eHi[i] = h.x, eHi[T + i] = h.y, eHi[2 * T + i] = h.z;
    eCen[i] = 0.5f * l.x + 0.5f * h.x, eCen[T + i] = 0.5f * l.y + 0.5f * h.y, eCen[2 * T + i] = 0.5f * l.z + 0.5f * h.z;
    eTid[i] = t;
  }
  segS[i] = 0, segE[i] = (uint16_t)n, segNode[i] = 0, segDep[i] = (uint16_t)it.depth;
  if (i == 0) nodeCount = 1u, maxDep = it.depth;
  __syncthreads();

  auto gather_boxes = [&](bool) {
    // sc[c][i] = box component c of the element that sits at sorted position i (key payload = its home position)
    const uint32_t h = i < n ? (uint32_t)(key[i] & 1023u) : 0u;
    for (int c = 0; c < 3; ++c) sc[c * T + i] = i < n ? eLo[c * T + h] : inf, sc[(3 + c) * T + i] = i < n ? eHi[c * T + h] : -inf;
  };
  auto scan_boxes = [&](bool suffix) {
    // segmented inclusive scan (union of boxes) along the positions, never across a segment border
    for (uint32_t d = 1; d < (uint32_t)T; d <<= 1) {
      float v[6];
      const bool take = suffix ? (i + d < segE[i]) : (i >= d + segS[i]);
      const uint32_t j = suffix ? i + d : i - d;
      for (int c = 0; c < 6; ++c) v[c] = sc[c * T + i];
      if (take)
        for (int c = 0; c < 3; ++c) v[c] = fminf(v[c], sc[c * T + j]), v[3 + c] = fmaxf(v[3 + c], sc[(3 + c) * T + j]);
      __syncthreads();
      for (int c = 0; c < 6; ++c) sc[c * T + i] = v[c];
      __syncthreads();
    }
  };
  auto area_at = [&](uint32_t j) {
    const float dx = sc[3 * T + j] - sc[j], dy = sc[4 * T + j] - sc[T + j], dz = sc[5 * T + j] - sc[2 * T + j];
    return dx < 0.f ? 0.f : dx * dy + dy * dz + dz * dx;
  };

  for (;;) {
    // ---- anything left to split?
    if (i == 0) anySplit = 0u;
    best[i] = ~0ull;
    __syncthreads();
    const bool big = i < n && (uint32_t)(segE[i] - segS[i]) > leafMax;
    if (big && i == segS[i]) anySplit = 1u;
    __syncthreads();
    if (!anySplit) break;
    // ---- the three sweeps
    for (int a = 0; a < 3; ++a) {
      uint32_t ck = 0;
      if (i < n) {
        const int fk = fkey(eCen[a * T + i]);
        ck = (uint32_t)fk ^ 0x80000000u;  // order-preserving unsigned
      }
      key[i] = i < n ? ((unsigned long long)segS[i] << 42) | ((unsigned long long)ck << 10) | i : ~0ull;
      __syncthreads();
      for (uint32_t k = 2; k <= (uint32_t)T; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const unsigned long long ka = key[i], kb = key[x];
            const bool up = (i & k) == 0;
            if ((ka > kb) == up) key[i] = kb, key[x] = ka;
          }
          __syncthreads();
        }
      if (i < n) rnk[a * T + (uint32_t)(key[i] & 1023u)] = (uint16_t)i;
      // suffix areas: sufA[j] = area of the box of sorted positions [j, segE)
      gather_boxes(true);
      __syncthreads();
      scan_boxes(true);
      sufA[i] = area_at(i);
      __syncthreads();
      gather_boxes(false);
      __syncthreads();
      scan_boxes(false);
      // the split AFTER sorted position i: left = [segS, i + 1), right = [i + 1, segE)
      if (big && i + 1u < segE[i]) {
        const uint32_t s0 = segS[i], nl = i + 1u - s0, nr = segE[i] - i - 1u;
        const int rem = depthCap - (int)segDep[i] - 1;
        const unsigned long long maxSide = rem >= 31 ? ~0ull : (unsigned long long)leafMax << (rem < 0 ? 0 : rem);
        if (nl <= maxSide && nr <= maxSide) {
          const float cost = area_at(i) * (float)((nl + leafMax - 1u) / leafMax) + sufA[i + 1u] * (float)((nr + leafMax - 1u) / leafMax);
          if (cost == cost && cost >= 0.f)
            atomicMin(&best[s0], ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)a << 16) | (i + 1u));
        }
      }
      __syncthreads();
    }
    // ---- decisions: (axis, split position) per segment; no admissible split (depth budget): the median of the
    // current order always fits it
    uint32_t ax = 0, kpos = 0;
    if (big) {
      const unsigned long long bb = best[segS[i]];
      if (bb == ~0ull) ax = 3u, kpos = segS[i] + (uint32_t)(segE[i] - segS[i]) / 2u;
      else ax = (uint32_t)(bb >> 16) & 3u, kpos = (uint32_t)bb & 0xffffu;
    }
    // ---- every triangle to its rank along the chosen axis (ax == 3: stays)
    const uint32_t np = (big && ax < 3u) ? rnk[ax * T + i] : i;
    float m[9];
    uint32_t mt = 0;
    uint16_t ms = 0, me = 0, mn = 0, md = 0;
    if (i < n) {
      for (int c = 0; c < 3; ++c) m[c] = eLo[c * T + i], m[3 + c] = eHi[c * T + i], m[6 + c] = eCen[c * T + i];
      mt = eTid[i], ms = segS[i], me = segE[i], mn = segNode[i], md = segDep[i];
    }
    __syncthreads();
    if (i < n) {
      for (int c = 0; c < 3; ++c) eLo[c * T + np] = m[c], eHi[c * T + np] = m[3 + c], eCen[c * T + np] = m[6 + c];
      eTid[np] = mt;
      // the new segment of this triangle, its parent's node and which child it is (in segNode's high bit for now)
      if (big) {
        const bool right = np >= kpos;
        segS[np] = right ? (uint16_t)kpos : ms, segE[np] = right ? me : (uint16_t)kpos;
        segDep[np] = (uint16_t)(md + 1u);
        segNode[np] = (uint16_t)(mn | (right ? 0x8000u : 0u));  // parent's node, child slot
      } else {
        segS[np] = ms, segE[np] = me, segDep[np] = md, segNode[np] = 0xffffu;  // a finished leaf: nothing to emit
      }
    }
    __syncthreads();
    // ---- boxes of the new segments: prefix scan in position order, the value at the segment's last position
    for (int c = 0; c < 3; ++c) sc[c * T + i] = i < n ? eLo[c * T + i] : inf, sc[(3 + c) * T + i] = i < n ? eHi[c * T + i] : -inf;
    __syncthreads();
    scan_boxes(false);
    // ---- number the new inner nodes (segment starts of segments that still exceed a leaf), in position order
    const bool fresh = i < n && segNode[i] != 0xffffu;            // belongs to a segment created by this level
    const bool starts = fresh && i == segS[i];
    const bool inner = starts && (uint32_t)(segE[i] - segS[i]) > leafMax;
    scan16[i] = inner ? 1u : 0u;
    __syncthreads();
    for (uint32_t d = 1; d < (uint32_t)T; d <<= 1) {
      const uint16_t v = i >= d ? scan16[i - d] : 0;
      __syncthreads();
      scan16[i] = (uint16_t)(scan16[i] + v);
      __syncthreads();
    }
    const uint32_t base = nodeCount;
    __syncthreads();
    // ---- the parents' records: child box + child ref, written by the first position of each new segment
    if (starts) {
      const uint32_t s0 = segS[i], e0 = segE[i], last = e0 - 1u;
      const uint32_t parent = segNode[i] & 0x7fffu, slot = segNode[i] >> 15;
      int32_t ref;
      if (inner) ref = (int32_t)(base + scan16[i] - 1u);
      else ref = ~(int32_t)(((it.b + s0) << 3) | (e0 - s0 - 1u));
      float* rec = reinterpret_cast<float*>(out + 4 * (size_t)parent);  // rtbvh::Node: lo0 hi0 lo1 hi1 child[2] pad[2]
      for (int c = 0; c < 3; ++c) rec[6 * slot + c] = sc[c * T + last], rec[6 * slot + 3 + c] = sc[(3 + c) * T + last];
      reinterpret_cast<int32_t*>(rec)[12 + slot] = ref;
      if (slot == 0) rec[14] = 0.f, rec[15] = 0.f;
      atomicMax(&maxDep, (uint32_t)segDep[i]);
    }
    __syncthreads();
    // the new segments carry their own node index from now on
    if (fresh) {
      const uint32_t s0 = segS[i];
      const bool in = (uint32_t)(segE[i] - s0) > leafMax;
      // (the index was computed by the segment's first position: read it back through the scan)
      segNode[i] = in ? (uint16_t)(base + scan16[s0] - 1u) : (uint16_t)0xffffu;
    }
    if (i == T - 1) nodeCount = base + scan16[T - 1];
    __syncthreads();
  }
  // ---- leaves: ascending triangle index inside a leaf, then the range's new order
  if (i < n && i == segS[i]) {
    const uint32_t s0 = segS[i], e0 = segE[i];
    for (uint32_t x = s0 + 1u; x < e0; ++x) {
      const uint32_t v = eTid[x];
      uint32_t y = x;
      while (y > s0 && eTid[y - 1u] > v) eTid[y] = eTid[y - 1u], --y;
      eTid[y] = v;
    }
  }
  __syncthreads();
  if (i < n) order[it.b + i] = eTid[i];
  if (i == 0) {
    cntOut[blockIdx.x] = nodeCount;
    atomicMax(heightOut, maxDep);
  }
}

constexpr size_t subtree_lds_bytes(int T) { return (size_t)T * (8 + 8 + 36 + 4 + 24 + 4 + 6 + 12); }

// scratch -> final arrays: subtree i's nodes go to [base[i], base[i] + cnt[i]), inner refs shifted, boxes padded
// and packed; the top node that refers to the subtree gets its root's index.
__global__ void k_sub_relocate(const SubItem* __restrict__ items, const uint32_t* __restrict__ scratchOff, const uint32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ finalOff, uint32_t nTop, const float4* __restrict__ scratch, float pad,
                               float boxScale, uint4* __restrict__ nodes16, float4* __restrict__ nodesF) {
  const SubItem it = items[blockIdx.x];
  const uint32_t base = nTop + finalOff[blockIdx.x], c = cnt[blockIdx.x];
  const float4* src = scratch + 4 * (size_t)scratchOff[blockIdx.x];
  for (uint32_t k = threadIdx.x; k < c; k += blockDim.x) {
    float r[16];
    for (int q = 0; q < 4; ++q) {
      const float4 v = src[4 * (size_t)k + q];
      r[4 * q] = v.x, r[4 * q + 1] = v.y, r[4 * q + 2] = v.z, r[4 * q + 3] = v.w;
    }
    int32_t child[2] = {__float_as_int(r[12]), __float_as_int(r[13])};
    for (int q = 0; q < 2; ++q)
      if (child[q] >= 0) child[q] += (int32_t)base;
    Box3 B0{r[0] - pad, r[1] - pad, r[2] - pad, r[3] + pad, r[4] + pad, r[5] + pad};
    Box3 B1{r[6] - pad, r[7] - pad, r[8] - pad, r[9] + pad, r[10] + pad, r[11] + pad};
    if (half_area(B1) < half_area(B0)) {  // child 0 = the smaller box (see k_level_emit)
      const Box3 t = B0;
      B0 = B1, B1 = t;
      const int32_t c0 = child[0];
      child[0] = child[1], child[1] = c0;
    }
    const size_t i = base + k;
    nodesF[4 * i + 0] = make_float4(B0.lx, B0.ly, B0.lz, B0.hx);
    nodesF[4 * i + 1] = make_float4(B0.hy, B0.hz, B1.lx, B1.ly);
    nodesF[4 * i + 2] = make_float4(B1.lz, B1.hx, B1.hy, B1.hz);
    nodesF[4 * i + 3] = make_float4(__int_as_float(child[0]), __int_as_float(child[1]), 0.f, 0.f);
    const float s = boxScale;
    const uint32_t h[12] = {half_directed(B0.lx * s, false), half_directed(B0.hx * s, true), half_directed(B0.ly * s, false),
                            half_directed(B0.hy * s, true),  half_directed(B0.lz * s, false), half_directed(B0.hz * s, true),
                            half_directed(B1.lx * s, false), half_directed(B1.hx * s, true), half_directed(B1.ly * s, false),
                            half_directed(B1.hy * s, true),  half_directed(B1.lz * s, false), half_directed(B1.hz * s, true)};
    nodes16[2 * i + 0] = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
    nodes16[2 * i + 1] = make_uint4(h[8] | h[9] << 16, h[10] | h[11] << 16, packed_ref(child[0]), packed_ref(child[1]));
  }
  if (threadIdx.x == 0 && it.parent != ~0u) {
    // the referring top node: float form child[slot] (word 12 + slot) and packed form (word 6 + slot of the 8)
    reinterpret_cast<int32_t*>(nodesF)[16 * (size_t)it.parent + 12 + it.slot] = (int32_t)base;
    reinterpret_cast<uint32_t*>(nodes16)[8 * (size_t)it.parent + 6 + it.slot] = base * 32u;
  }
}
__global__ void k_sub_sizes(const SubItem* __restrict__ items, uint32_t n, uint32_t* __restrict__ sizes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sizes[i] = items[i].e - items[i].b - 1u;  // a binary tree over m triangles has at most m - 1 inner nodes
}

// ---------------------------------------------------------------- tree rotations on the device (round 4)
// bvh_build.cpp's Rotator as kernels: at every inner node N with children (L, R), swapping R with a child of L (and the
// mirror cases), or a child of L with a child of R, is applied when it shrinks the summed surface area of the boxes —
// children before parents, never pushing a subtree below the depth cap.  The recursion becomes one launch per DEPTH,
// bottom-up: the nodes of one depth have disjoint subtrees and a rotation only touches its node, the node's two children
// and their child slots, so they are independent; moved subtrees change depth only BELOW the rotated node, which the
// levels still to come never look at (they read child records and heights).  Per pass: parents -> depths -> heights ->
// rotations; afterwards every record is re-packed (child 0 = the smaller box again).
using RNode = rtbvh::Node;  // 64 B: lo0 hi0 lo1 hi1 child[2] pad[2] — the layout of nodesF

__global__ void k_rot_parents(const RNode* __restrict__ nodes, uint32_t n, uint32_t* __restrict__ parent) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0) parent[0] = ~0u;
  for (int c = 0; c < 2; ++c)
    if (nodes[i].child[c] >= 0) parent[nodes[i].child[c]] = i;
}
__global__ void k_rot_depths(const uint32_t* __restrict__ parent, uint32_t n, uint8_t* __restrict__ depth, uint32_t* __restrict__ maxDepth) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t d = 0;
  for (uint32_t a = parent[i]; a != ~0u && d < 64u; a = parent[a]) ++d;
  depth[i] = (uint8_t)d;
  atomicMax(maxDepth, d);
}
__device__ __forceinline__ int rot_height(const uint8_t* height, int32_t ref) { return ref < 0 ? 0 : (int)height[ref]; }
__global__ void k_rot_heights(const RNode* __restrict__ nodes, uint32_t n, const uint8_t* __restrict__ depth, uint32_t d,
                              uint8_t* __restrict__ height) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || depth[i] != d) return;
  height[i] = (uint8_t)(1 + max(rot_height(height, nodes[i].child[0]), rot_height(height, nodes[i].child[1])));
}
__device__ __forceinline__ Box3 rot_child(const RNode& n, int i) {
  return i ? Box3{n.lo1[0], n.lo1[1], n.lo1[2], n.hi1[0], n.hi1[1], n.hi1[2]} : Box3{n.lo0[0], n.lo0[1], n.lo0[2], n.hi0[0], n.hi0[1], n.hi0[2]};
}
__device__ __forceinline__ void rot_set(RNode& n, int i, int32_t ref, const Box3& b) {
  if (i) n.lo1[0] = b.lx, n.lo1[1] = b.ly, n.lo1[2] = b.lz, n.hi1[0] = b.hx, n.hi1[1] = b.hy, n.hi1[2] = b.hz;
  else n.lo0[0] = b.lx, n.lo0[1] = b.ly, n.lo0[2] = b.lz, n.hi0[0] = b.hx, n.hi0[1] = b.hy, n.hi0[2] = b.hz;
  n.child[i] = ref;
}
__device__ __forceinline__ Box3 rot_union(const Box3& a, const Box3& b) {
  return Box3{fminf(a.lx, b.lx), fminf(a.ly, b.ly), fminf(a.lz, b.lz), fmaxf(a.hx, b.hx), fmaxf(a.hy, b.hy), fmaxf(a.hz, b.hz)};
}
__global__ void k_rot_level(RNode* __restrict__ nodes, uint32_t n, const uint8_t* __restrict__ depth, uint32_t d, int depthCap,
                            uint8_t* __restrict__ height, uint32_t* __restrict__ swaps) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n || depth[idx] != d) return;
  RNode N = nodes[idx];
  float bestDelta = -1e-7f * (half_area(rot_child(N, 0)) + half_area(rot_child(N, 1)));
  int bestS = -1, bestG = -1;
  for (int s = 0; s < 2; ++s) {
    const int32_t l = N.child[s], r = N.child[1 - s];
    if (l < 0) continue;
    if ((int)d + 2 + rot_height(height, r) > depthCap) continue;  // r would sit one level deeper
    const Box3 rb = rot_child(N, 1 - s);
    const float oldArea = half_area(rot_child(N, s));
    const RNode L = nodes[l];
    for (int g = 0; g < 2; ++g) {  // grandchild g goes up, its sibling stays with r
      const float delta = half_area(rot_union(rot_child(L, 1 - g), rb)) - oldArea;
      if (delta < bestDelta) bestDelta = delta, bestS = s, bestG = g;
    }
  }
  int bestX = -1;
  if (N.child[0] >= 0 && N.child[1] >= 0) {  // grandchild <-> grandchild across the two sides (no depth change)
    const RNode L = nodes[N.child[0]], R = nodes[N.child[1]];
    const float oldArea = half_area(rot_child(N, 0)) + half_area(rot_child(N, 1));
    for (int x = 0; x < 2; ++x) {  // L.child[0] swaps with R.child[x]
      const Box3 lb = rot_union(rot_child(R, x), rot_child(L, 1)), rb = rot_union(rot_child(L, 0), rot_child(R, 1 - x));
      const float delta = half_area(lb) + half_area(rb) - oldArea;
      if (delta < bestDelta) bestDelta = delta, bestX = x, bestS = -1;
    }
  }
  if (bestX >= 0) {
    const int32_t li = N.child[0], ri = N.child[1];
    RNode L = nodes[li], R = nodes[ri];
    const Box3 a0 = rot_child(L, 0), bx = rot_child(R, bestX);
    const int32_t ra = L.child[0], rb = R.child[bestX];
    rot_set(L, 0, rb, bx);
    rot_set(R, bestX, ra, a0);
    rot_set(N, 0, li, rot_union(rot_child(L, 0), rot_child(L, 1)));
    rot_set(N, 1, ri, rot_union(rot_child(R, 0), rot_child(R, 1)));
    nodes[li] = L, nodes[ri] = R;
    height[li] = (uint8_t)(1 + max(rot_height(height, L.child[0]), rot_height(height, L.child[1])));
    height[ri] = (uint8_t)(1 + max(rot_height(height, R.child[0]), rot_height(height, R.child[1])));
    atomicAdd(swaps, 1u);
  } else if (bestS >= 0) {
    const int sidx = bestS, g = bestG;
    const int32_t l = N.child[sidx], r = N.child[1 - sidx];
    RNode L = nodes[l];
    const Box3 rb = rot_child(N, 1 - sidx), gb = rot_child(L, g);
    const int32_t gref = L.child[g];
    rot_set(L, g, r, rb);
    rot_set(N, 1 - sidx, gref, gb);
    rot_set(N, sidx, l, rot_union(rot_child(L, 0), rot_child(L, 1)));
    nodes[l] = L;
    height[l] = (uint8_t)(1 + max(rot_height(height, L.child[0]), rot_height(height, L.child[1])));
    atomicAdd(swaps, 1u);
  }
  nodes[idx] = N;
  height[idx] = (uint8_t)(1 + max(rot_height(height, N.child[0]), rot_height(height, N.child[1])));
}
// float records -> packed records (child 0 = the smaller box: any-hit rays of the big-scene kernels enter it first)
__global__ void k_rot_pack(RNode* __restrict__ nodes, uint32_t n, float boxScale, uint4* __restrict__ nodes16) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RNode N = nodes[i];
  Box3 B0 = rot_child(N, 0), B1 = rot_child(N, 1);
  if (half_area(B1) < half_area(B0)) {
    const Box3 t = B0;
    B0 = B1, B1 = t;
    const int32_t c0 = N.child[0];
    N.child[0] = N.child[1], N.child[1] = c0;
    rot_set(N, 0, N.child[0], B0), rot_set(N, 1, N.child[1], B1);
    nodes[i] = N;
  }
  const float s = boxScale;
  const uint32_t h[12] = {half_directed(B0.lx * s, false), half_directed(B0.hx * s, true), half_directed(B0.ly * s, false),
                          half_directed(B0.hy * s, true),  half_directed(B0.lz * s, false), half_directed(B0.hz * s, true),
                          half_directed(B1.lx * s, false), half_directed(B1.hx * s, true), half_directed(B1.ly * s, false),
                          half_directed(B1.hy * s, true),  half_directed(B1.lz * s, false), half_directed(B1.hz * s, true)};
  nodes16[2 * (size_t)i + 0] = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
  nodes16[2 * (size_t)i + 1] = make_uint4(h[8] | h[9] << 16, h[10] | h[11] << 16, packed_ref(N.child[0]), packed_ref(N.child[1]));
}

// Pre-order renumbering after the rotations (what the host builder's relayout leaves below its top, and what the
// traversal's locality was tuned on): subtree sizes bottom-up, new indices top-down — a node's first inner child follows
// it, the second follows the first one's subtree —, then a scatter with the refs rewritten.
__global__ void k_rot_sizes(const RNode* __restrict__ nodes, uint32_t n, const uint8_t* __restrict__ depth, uint32_t d,
                            uint32_t* __restrict__ size) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || depth[i] != d) return;
  const int32_t c0 = nodes[i].child[0], c1 = nodes[i].child[1];
  size[i] = 1u + (c0 >= 0 ? size[c0] : 0u) + (c1 >= 0 ? size[c1] : 0u);
}
__global__ void k_rot_number(const RNode* __restrict__ nodes, uint32_t n, const uint8_t* __restrict__ depth, uint32_t d,
                             const uint32_t* __restrict__ size, uint32_t* __restrict__ newIdx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || depth[i] != d) return;
  if (d == 0u) newIdx[i] = 0u;
  const uint32_t me = newIdx[i];
  const int32_t c0 = nodes[i].child[0], c1 = nodes[i].child[1];
  if (c0 >= 0) newIdx[c0] = me + 1u;
  if (c1 >= 0) newIdx[c1] = me + 1u + (c0 >= 0 ? size[c0] : 0u);
}
__global__ void k_rot_scatter(const RNode* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ newIdx, RNode* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RNode N = nodes[i];
  for (int c = 0; c < 2; ++c)
    if (N.child[c] >= 0) N.child[c] = (int32_t)newIdx[N.child[c]];
  out[newIdx[i]] = N;
}

// `passes` rotation passes over the n float records at nodesF (0 = none), then the packed records; *maxDepthOut = the deepest
// leaf level afterwards.  RT_BVH_GPU_ROT overrides the pass count.
static hipError_t rotate_and_pack(float4* nodesF, uint4* nodes16, uint32_t n, int depthCap, float boxScale, int passes,
                                  uint32_t* maxDepthOut, hipStream_t stream) {
  static const int rotEnv = getenv("RT_BVH_GPU_ROT") ? atoi(getenv("RT_BVH_GPU_ROT")) : -1;
  if (rotEnv >= 0) passes = rotEnv;
  if (passes <= 0 || n < 2) return hipSuccess;
  RNode* nodes = reinterpret_cast<RNode*>(nodesF);
  uint32_t *parent = nullptr, *scal = nullptr;  // scal: [0] max depth, [1] swaps
  uint8_t *depth = nullptr, *height = nullptr;
  hipError_t e = hipMalloc((void**)&parent, (size_t)n * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&scal, 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&depth, n);
  if (e == hipSuccess) e = hipMalloc((void**)&height, n);
  const dim3 blk(256), grd((n + 255) / 256);
  uint32_t rootHeight = 0;
  for (int p = 0; p < passes && e == hipSuccess; ++p) {
    e = hipMemsetAsync(scal, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_rot_parents, grd, blk, 0, stream, nodes, n, parent);
    hipLaunchKernelGGL(k_rot_depths, grd, blk, 0, stream, parent, n, depth, scal);
    uint32_t h2[2] = {0, 0};
    e = hipMemcpyAsync(h2, scal, sizeof h2, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) break;
    const uint32_t maxD = h2[0];
    for (int d = (int)maxD; d >= 0; --d) hipLaunchKernelGGL(k_rot_heights, grd, blk, 0, stream, nodes, n, depth, (uint32_t)d, height);
    for (int d = (int)maxD; d >= 0; --d) hipLaunchKernelGGL(k_rot_level, grd, blk, 0, stream, nodes, n, depth, (uint32_t)d, depthCap, height, scal + 1);
    uint8_t rh = 0;
    e = hipMemcpyAsync(h2, scal, sizeof h2, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&rh, height, 1, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    rootHeight = rh;
    if (getenv("RT_BVH_VERBOSE")) fprintf(stderr, "device rotation pass %d: %u swaps, tree depth %u\n", p, h2[1], rootHeight);
    if (e == hipSuccess && h2[1] == 0) break;
  }
  static const bool preorder = !getenv("RT_BVH_GPU_PREORDER") || atoi(getenv("RT_BVH_GPU_PREORDER")) != 0;
  RNode* tmpNodes = nullptr;
  uint32_t* size = nullptr;
  if (e == hipSuccess && preorder) {
    e = hipMalloc((void**)&tmpNodes, (size_t)n * sizeof(RNode));
    if (e == hipSuccess) e = hipMalloc((void**)&size, (size_t)n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(scal, 0, 2 * sizeof(uint32_t), stream);
    uint32_t h2[2] = {0, 0};
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_rot_parents, grd, blk, 0, stream, nodes, n, parent);
      hipLaunchKernelGGL(k_rot_depths, grd, blk, 0, stream, parent, n, depth, scal);
      e = hipMemcpyAsync(h2, scal, sizeof h2, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e == hipSuccess) {
      const int maxD = (int)h2[0];
      uint32_t* newIdx = parent;  // (the parents are not needed any more)
      for (int d = maxD; d >= 0; --d) hipLaunchKernelGGL(k_rot_sizes, grd, blk, 0, stream, nodes, n, depth, (uint32_t)d, size);
      for (int d = 0; d <= maxD; ++d) hipLaunchKernelGGL(k_rot_number, grd, blk, 0, stream, nodes, n, depth, (uint32_t)d, size, newIdx);
      hipLaunchKernelGGL(k_rot_scatter, grd, blk, 0, stream, nodes, n, newIdx, tmpNodes);
      e = hipMemcpyAsync(nodes, tmpNodes, (size_t)n * sizeof(RNode), hipMemcpyDeviceToDevice, stream);
    }
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_rot_pack, grd, blk, 0, stream, nodes, n, boxScale, nodes16);
    e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipGetLastError();
  }
  for (void* q : {(void*)parent, (void*)scal, (void*)depth, (void*)height, (void*)tmpNodes, (void*)size})
    if (q) (void)hipFree(q);
  if (e == hipSuccess && rootHeight && maxDepthOut) *maxDepthOut = rootHeight;
  return e;
}

#define GB_TRY(expr)            \
  do {                          \
    hipError_t e_ = (expr);     \
    if (e_ != hipSuccess) {     \
      cleanup();                \
      return e_;                \
    }                           \
  } while (0)

}  // namespace

// Builds nodes16 / nodesF / tris / trisRef on the current device.  The output arrays are
// hipMalloc'ed here and owned by the caller.  Returns hipSuccess and fills `out`.
hipError_t gpu_bvh_build(const float* dVpos, const uint4* dTriShade, uint32_t n, const rtbvh::ScenePlan& P, GpuBvh* out,
                         hipStream_t stream) {
  *out = GpuBvh{};
  const uint32_t leafMax = P.leafMax;
  uint32_t N2 = 1;
  while (N2 < n) N2 <<= 1;
  float4 *lo = nullptr, *hi = nullptr, *segLo = nullptr, *segHi = nullptr, *nodesF = nullptr, *tris = nullptr, *trisRef = nullptr;
  uint4* nodes16 = nullptr;
  int* cb = nullptr;
  uint64_t *keys = nullptr, *keys2 = nullptr;
  uint32_t *vals = nullptr, *order = nullptr, *splitPos = nullptr, *innerCnt = nullptr, *innerOff = nullptr;
  WorkItem *itemsA = nullptr, *itemsB = nullptr;
  SubItem* subs = nullptr;
  uint32_t *subCnt = nullptr, *subOff = nullptr, *subSizes = nullptr, *scratchOff = nullptr, *subNodes = nullptr, *finalOff = nullptr,
           *height = nullptr;
  float4* scratch = nullptr;
  void* tmp = nullptr;
  bool keepOutputs = false;
  static const uint32_t subMax = getenv("RT_BVH_GPU_SUB") ? (uint32_t)atoi(getenv("RT_BVH_GPU_SUB")) : kSubMax;  // 0: Morton cuts all the way down
  // the top of the tree (ranges above subMax): the cheapest cut of the static Morton order.  (Round 3 also had the host's
  // binned SAH as a device kernel here: worse trees under 1,024-triangle subtrees and 3 x the time,
  // profiles/r03_device_bvh_top_variants.txt; round 4's hybrid builder — gpu_bvh_build_over_top below — takes the host's
  // OWN top instead.)
  auto cleanup = [&]() {
    for (void* p : {(void*)lo, (void*)hi, (void*)segLo, (void*)segHi, (void*)cb, (void*)keys, (void*)keys2, (void*)vals, (void*)order,
                    (void*)splitPos, (void*)innerCnt, (void*)innerOff, (void*)itemsA, (void*)itemsB, (void*)subs, (void*)subCnt,
                    (void*)subOff, (void*)subSizes, (void*)scratchOff, (void*)subNodes, (void*)finalOff, (void*)height, (void*)scratch,
                    tmp})
      if (p) (void)hipFree(p);
    if (!keepOutputs)
      for (void* p : {(void*)nodes16, (void*)nodesF, (void*)tris, (void*)trisRef})
        if (p) (void)hipFree(p);
  };
  const uint32_t maxNodes = n;  // a binary tree over n > leafMax triangles with >= 1 per leaf has < n inner nodes
  GB_TRY(hipMalloc((void**)&lo, (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&hi, (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&segLo, 2 * (size_t)N2 * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&segHi, 2 * (size_t)N2 * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&cb, 6 * sizeof(int)));
  GB_TRY(hipMalloc((void**)&keys, (size_t)n * sizeof(uint64_t)));
  GB_TRY(hipMalloc((void**)&keys2, (size_t)n * sizeof(uint64_t)));
  GB_TRY(hipMalloc((void**)&vals, (size_t)n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&order, (size_t)n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&splitPos, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&innerCnt, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&innerOff, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&itemsA, (size_t)maxNodes * sizeof(WorkItem)));
  GB_TRY(hipMalloc((void**)&itemsB, (size_t)maxNodes * sizeof(WorkItem)));
  GB_TRY(hipMalloc((void**)&subs, (size_t)maxNodes * sizeof(SubItem)));
  GB_TRY(hipMalloc((void**)&subCnt, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&subOff, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&subSizes, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&scratchOff, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&subNodes, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&finalOff, (size_t)maxNodes * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&height, sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&scratch, 4 * (size_t)maxNodes * sizeof(float4)));
  GB_TRY(hipMemsetAsync(height, 0, sizeof(uint32_t), stream));
  GB_TRY(hipMalloc((void**)&nodes16, 2 * (size_t)maxNodes * sizeof(uint4)));
  GB_TRY(hipMalloc((void**)&nodesF, 4 * (size_t)maxNodes * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&tris, 3 * (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&trisRef, 3 * (size_t)n * sizeof(float4)));
  size_t sortBytes = 0, scanBytes = 0;
  GB_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, keys, keys2, vals, order, n, 0, 63, stream));
  GB_TRY(rocprim::exclusive_scan(nullptr, scanBytes, innerCnt, innerOff, 0u, (size_t)maxNodes, rocprim::plus<uint32_t>(), stream));
  const size_t tmpBytes = sortBytes > scanBytes ? sortBytes : scanBytes;
  GB_TRY(hipMalloc(&tmp, tmpBytes ? tmpBytes : 16));

  const dim3 blk(256), grdN((n + 255) / 256);
  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, cb);
  hipLaunchKernelGGL(k_tri_boxes, grdN, blk, 0, stream, dVpos, dTriShade, n, lo, hi, cb);
  hipLaunchKernelGGL(k_morton, grdN, blk, 0, stream, lo, hi, n, cb, keys, vals);
  GB_TRY(rocprim::radix_sort_pairs(tmp, sortBytes, keys, keys2, vals, order, n, 0, 63, stream));
  hipLaunchKernelGGL(k_seg_leaves, dim3((N2 + 255) / 256), blk, 0, stream, lo, hi, order, n, N2, segLo, segHi);
  for (uint32_t cnt = N2 / 2; cnt >= 1; cnt >>= 1)  // level with `cnt` nodes starts at index cnt
    hipLaunchKernelGGL(k_seg_level, dim3((cnt + 255) / 256), blk, 0, stream, cnt, cnt, segLo, segHi);
  // top-down, one level at a time; ranges of <= subMax triangles become items of the exact builder
  const WorkItem root{0u, n};
  GB_TRY(hipMemcpyAsync(itemsA, &root, sizeof root, hipMemcpyHostToDevice, stream));
  uint32_t count = 1, levelBase = 0, depth = 0, maxDepth = 0, nSub = 0;
  if (n <= subMax) {  // the whole scene is one exact subtree
    const SubItem whole{0u, n, 0u, ~0u, 0u};
    GB_TRY(hipMemcpyAsync(subs, &whole, sizeof whole, hipMemcpyHostToDevice, stream));
    nSub = 1, count = 0;
  }
  WorkItem *cur = itemsA, *nxt = itemsB;
  while (count) {
    if ((int)depth >= rtbvh::kMaxDepth - 1 || levelBase + count > maxNodes) {
      cleanup();
      return hipErrorInvalidValue;  // (cannot happen: the depth budget is enforced by the split choice)
    }
    hipLaunchKernelGGL(k_level_split, dim3((count + 3) / 4), dim3(256), 0, stream, cur, count, depth, P.depthCap, leafMax, segLo,
                       segHi, N2, keys2, splitPos, innerCnt, subCnt, subMax);
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, innerCnt, innerOff, 0u, (size_t)count, rocprim::plus<uint32_t>(), stream));
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, subCnt, subOff, 0u, (size_t)count, rocprim::plus<uint32_t>(), stream));
    const uint32_t nextBase = levelBase + count;
    hipLaunchKernelGGL(k_level_emit, dim3((count + 255) / 256), blk, 0, stream, cur, count, levelBase, nextBase, splitPos, innerOff,
                       leafMax, segLo, segHi, N2, P.pad, P.boxScale, nodes16, nodesF, nxt, subOff, nSub, subs, subMax, depth);
    uint32_t lastOff = 0, lastCnt = 0, lastSubOff = 0, lastSubCnt = 0;
    GB_TRY(hipMemcpyAsync(&lastOff, innerOff + (count - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&lastCnt, innerCnt + (count - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&lastSubOff, subOff + (count - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&lastSubCnt, subCnt + (count - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    maxDepth = depth + 1;  // leaves hang one level below the deepest inner level
    levelBase = nextBase;
    count = lastOff + lastCnt;
    nSub += lastSubOff + lastSubCnt;
    ++depth;
    WorkItem* t = cur;
    cur = nxt, nxt = t;
  }
  uint32_t nTotal = levelBase;
  if (nSub) {
    // the exact subtrees: scratch blocks of (triangles - 1) node slots each, built one workgroup per range ...
    hipLaunchKernelGGL(k_sub_sizes, dim3((nSub + 255) / 256), blk, 0, stream, subs, nSub, subSizes);
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, subSizes, scratchOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    // (dynamic LDS beyond 64 KiB is granted per kernel AND per device — one process may build on several devices:
    // rt_group —, so the grant is tracked per device, as rt_kernels.hip allow_big_lds does)
    const size_t ldsBytes = subtree_lds_bytes((int)kSubMax);
    {
      static std::atomic<unsigned long long> ldsSet{0};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = -1;
      if (dev < 0 || !(ldsSet.load() & (1ull << dev))) {
        GB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_subtree<(int)kSubMax>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)ldsBytes));
        if (dev >= 0) ldsSet.fetch_or(1ull << dev);
      }
    }
    hipLaunchKernelGGL((k_subtree<(int)kSubMax>), dim3(nSub), dim3(kSubMax), ldsBytes, stream, subs, scratchOff, order, lo, hi, leafMax,
                       P.depthCap, scratch, subNodes, height);
    // ... then numbered behind the top in item order (deterministic) and packed
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, subNodes, finalOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_sub_relocate, dim3(nSub), blk, 0, stream, subs, scratchOff, subNodes, finalOff, levelBase, scratch, P.pad,
                       P.boxScale, nodes16, nodesF);
    uint32_t lastOff = 0, lastCnt = 0, h = 0;
    GB_TRY(hipMemcpyAsync(&lastOff, finalOff + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&lastCnt, subNodes + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&h, height, 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    nTotal = levelBase + lastOff + lastCnt;
    maxDepth = maxDepth > h ? maxDepth : h;
    if (nTotal > maxNodes) {
      cleanup();
      return hipErrorInvalidValue;
    }
  }
  // rotation passes over the whole tree (as the host builder's: 3 on big scenes, 8 on small ones)
  GB_TRY(rotate_and_pack(nodesF, nodes16, nTotal, P.depthCap, P.boxScale, n > 200000u ? 3 : 8, &maxDepth, stream));
  // triangle records in the FINAL leaf order (the exact builder has reordered its ranges), and in reference order
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, order, n, tris);
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, (const uint32_t*)nullptr, n, trisRef);
  GB_TRY(hipStreamSynchronize(stream));
  GB_TRY(hipGetLastError());
  keepOutputs = true;
  out->nodes16 = nodes16, out->nodesF = nodesF, out->tris = tris, out->trisRef = trisRef;
  out->n_nodes = nTotal, out->maxDepth = maxDepth;
  cleanup();
  return hipSuccess;
}

// The HYBRID build (rt_options.bvh_builder = RT_BVH_HYBRID): the host builder's own top (rtbvh::buildTop: its split
// choices down to parts of <= kSubMax triangles) and, below it, the exact subtrees of step 6 — one workgroup per part.
// The host spends most of a build in those bottom levels (sorts of every range of <= 4,096 triangles along four axes);
// the top is a few binned passes.  Same arrays out as gpu_bvh_build.
hipError_t gpu_bvh_build_over_top(const float* dVpos, const uint4* dTriShade, uint32_t n, const rtbvh::TopBuilt& top, GpuBvh* out,
                                  hipStream_t stream) {
  *out = GpuBvh{};
  const uint32_t nTop = (uint32_t)top.nodes.size(), nSub = (uint32_t)top.parts.size();
  if (n == 0 || nTop == 0 || top.order.size() != n) return hipErrorInvalidValue;
  for (const rtbvh::TopBuilt::Part& p : top.parts)
    if (p.e <= p.b || p.e - p.b > kSubMax || p.e > n || p.parent >= nTop || p.slot > 1u) return hipErrorInvalidValue;
  float4 *lo = nullptr, *hi = nullptr, *nodesF = nullptr, *tris = nullptr, *trisRef = nullptr, *scratch = nullptr;
  uint4* nodes16 = nullptr;
  int* cb = nullptr;
  uint32_t *order = nullptr, *subSizes = nullptr, *scratchOff = nullptr, *subNodes = nullptr, *finalOff = nullptr, *height = nullptr;
  SubItem* subs = nullptr;
  void* tmp = nullptr;
  bool keepOutputs = false;
  auto cleanup = [&]() {
    for (void* p : {(void*)lo, (void*)hi, (void*)cb, (void*)order, (void*)subSizes, (void*)scratchOff, (void*)subNodes, (void*)finalOff,
                    (void*)height, (void*)scratch, (void*)subs, tmp})
      if (p) (void)hipFree(p);
    if (!keepOutputs)
      for (void* p : {(void*)nodes16, (void*)nodesF, (void*)tris, (void*)trisRef})
        if (p) (void)hipFree(p);
  };
  const uint32_t maxNodes = n + nTop;
  const size_t nS = nSub ? nSub : 1u;
  GB_TRY(hipMalloc((void**)&lo, (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&hi, (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&cb, 6 * sizeof(int)));
  GB_TRY(hipMalloc((void**)&order, (size_t)n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&subs, nS * sizeof(SubItem)));
  GB_TRY(hipMalloc((void**)&subSizes, nS * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&scratchOff, nS * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&subNodes, nS * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&finalOff, nS * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&height, sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&scratch, 4 * (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&nodes16, 2 * (size_t)maxNodes * sizeof(uint4)));
  GB_TRY(hipMalloc((void**)&nodesF, 4 * (size_t)maxNodes * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&tris, 3 * (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&trisRef, 3 * (size_t)n * sizeof(float4)));
  size_t scanBytes = 0;
  GB_TRY(rocprim::exclusive_scan(nullptr, scanBytes, subSizes, scratchOff, 0u, nS, rocprim::plus<uint32_t>(), stream));
  GB_TRY(hipMalloc(&tmp, scanBytes ? scanBytes : 16));
  GB_TRY(hipMemsetAsync(height, 0, sizeof(uint32_t), stream));
  GB_TRY(hipMemcpyAsync(order, top.order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  // the top's records, float and packed (the packing of rtbvh::packNodes; part refs are patched in by k_sub_relocate)
  std::vector<rtbvh::Node16> top16(nTop);
  for (uint32_t i = 0; i < nTop; ++i) {
    const rtbvh::Node& nd = top.nodes[i];
    rtbvh::Node16& q = top16[i];
    for (int a = 0; a < 3; ++a) {
      q.box0[2 * a] = rtbvh::toHalfDirected(nd.lo0[a] * top.boxScale, false), q.box0[2 * a + 1] = rtbvh::toHalfDirected(nd.hi0[a] * top.boxScale, true);
      q.box1[2 * a] = rtbvh::toHalfDirected(nd.lo1[a] * top.boxScale, false), q.box1[2 * a + 1] = rtbvh::toHalfDirected(nd.hi1[a] * top.boxScale, true);
    }
    for (int c = 0; c < 2; ++c) {
      const int32_t ref = nd.child[c];
      if (ref >= 0) q.child[c] = ref * 32;
      else if ((~(uint32_t)ref) & rtbvh::kPartFlag) q.child[c] = 0;  // (patched)
      else q.child[c] = (int32_t)~(((~(uint32_t)ref) >> 3) * 48u | ((~(uint32_t)ref) & 7u));
    }
  }
  static_assert(sizeof(rtbvh::Node) == 4 * sizeof(float4) && sizeof(rtbvh::Node16) == 2 * sizeof(uint4), "node layouts");
  GB_TRY(hipMemcpyAsync(nodesF, top.nodes.data(), (size_t)nTop * sizeof(rtbvh::Node), hipMemcpyHostToDevice, stream));
  GB_TRY(hipMemcpyAsync(nodes16, top16.data(), (size_t)nTop * sizeof(rtbvh::Node16), hipMemcpyHostToDevice, stream));
  std::vector<SubItem> hs(nSub);
  for (uint32_t i = 0; i < nSub; ++i) hs[i] = SubItem{top.parts[i].b, top.parts[i].e, top.parts[i].depth, top.parts[i].parent, top.parts[i].slot};
  if (nSub) GB_TRY(hipMemcpyAsync(subs, hs.data(), (size_t)nSub * sizeof(SubItem), hipMemcpyHostToDevice, stream));
  const dim3 blk(256), grdN((n + 255) / 256);
  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, cb);
  hipLaunchKernelGGL(k_tri_boxes, grdN, blk, 0, stream, dVpos, dTriShade, n, lo, hi, cb);
  uint32_t nTotal = nTop, maxDepth = top.maxDepth;
  if (nSub) {
    hipLaunchKernelGGL(k_sub_sizes, dim3((nSub + 255) / 256), blk, 0, stream, subs, nSub, subSizes);
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, subSizes, scratchOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    const size_t ldsBytes = subtree_lds_bytes((int)kSubMax);
    {
      static std::atomic<unsigned long long> ldsSet{0};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = -1;
      if (dev < 0 || !(ldsSet.load() & (1ull << dev))) {
        GB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_subtree<(int)kSubMax>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)ldsBytes));
        if (dev >= 0) ldsSet.fetch_or(1ull << dev);
      }
    }
    hipLaunchKernelGGL((k_subtree<(int)kSubMax>), dim3(nSub), dim3(kSubMax), ldsBytes, stream, subs, scratchOff, order, lo, hi, top.leafMax,
                       top.depthCap, scratch, subNodes, height);
    GB_TRY(rocprim::exclusive_scan(tmp, scanBytes, subNodes, finalOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_sub_relocate, dim3(nSub), blk, 0, stream, subs, scratchOff, subNodes, finalOff, nTop, scratch, top.pad,
                       top.boxScale, nodes16, nodesF);
    uint32_t lastOff = 0, lastCnt = 0, h = 0;
    GB_TRY(hipMemcpyAsync(&lastOff, finalOff + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&lastCnt, subNodes + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&h, height, 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    nTotal = nTop + lastOff + lastCnt;
    maxDepth = maxDepth > h ? maxDepth : h;
    if (nTotal > maxNodes) {
      cleanup();
      return hipErrorInvalidValue;
    }
  }
  GB_TRY(rotate_and_pack(nodesF, nodes16, nTotal, top.depthCap, top.boxScale, n > 200000u ? 3 : 8, &maxDepth, stream));
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, order, n, tris);
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, (const uint32_t*)nullptr, n, trisRef);
  GB_TRY(hipStreamSynchronize(stream));
  GB_TRY(hipGetLastError());
  keepOutputs = true;
  out->nodes16 = nodes16, out->nodesF = nodesF, out->tris = tris, out->trisRef = trisRef;
  out->n_nodes = nTotal, out->maxDepth = maxDepth;
  cleanup();
  return hipSuccess;
}

}  // namespace rtk
