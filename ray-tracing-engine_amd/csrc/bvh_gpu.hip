// bvh_gpu.hip — the scene BVH built on the device.
//
// The reference has no usable BVH (source/BVH.h, AABB.cpp: dead code), so
// like the host builder (bvh_build.cpp) this is our own design; it emits the SAME arrays
// the traversal kernels read — 32-B binary16-packed nodes holding both child boxes, 48-B
// triangle records in leaf order, leaves of <= leafMax triangles, depth <= the plan's cap —
// so every exactness test runs unchanged on a device-built tree (RT_BVH_GPU=1).
//
//   1. per-triangle boxes; the size keys come from the host (rtbvh::planSceneExact);
//   2. THE TOP (ranges of more than kSubMax = 1,024 triangles): the host builder's split rules as kernels, one level of the
//      tree per round of launches (k_top_*): binned SAH over the three centroid axes and the size axis above 4,096 triangles
//      — bounds, bins of all four axes in one pass, the choice, a stable partition —, the exact sweep over the same four
//      axes below (one workgroup per range: bitonic sorts by (key, id), suffix / prefix box scans, every cut priced), the
//      depth budget, the host's tie rules.  Everything the host derives from min / max / counts or from sorts under a total
//      order comes out the same here, so the splits — and the tree — are the host builder's;
//   3. THE SUBTREES (round 3): one workgroup per range of <= kSubMax triangles builds the whole subtree with the exact
//      sweep — per level and axis a bitonic sort of the range's segments by centroid (LDS), segmented prefix / suffix box
//      scans, the SAH cost of every split position, an atomic min per segment — and reorders the range's triangles; the
//      subtrees are numbered behind the top (scan of their node counts);
//   4. the host builder's rotation passes as kernels (one launch per depth, bottom-up), a pre-order renumbering, the
//      packing with the smaller child box in slot 0; the triangle records in leaf order.
// The HYBRID build takes step 2 from the host (rtbvh::buildTop).
// (Rounds 2-3 had a Morton-order top here — radix sort, segment tree of boxes, 64 candidate cuts per range —: +4 ... +7 % node
// visits against the host tree; removed in round 4, code at commit 286f27d.)
// Result and cost (profiles/r04_builders.txt): the host builder's node visits per ray to the last digit on all
// four scenes, 1 M triangles in 21 ms (host: 170-180 ms).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <cstdio>
#include <cstring>  // (rocPRIM's headers use memset without including it)
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "bvh_build.h"
#include "rt_kernels.h"

namespace rtk {
namespace {

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

// triangle boxes (float4 lo, hi) in REFERENCE order
__global__ void k_tri_boxes(const float* __restrict__ vpos, const uint4* __restrict__ triShade, uint32_t n,
                            float4* __restrict__ lo, float4* __restrict__ hi) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint4 tv = triShade[t];
  float l[3], h[3];
  for (int a = 0; a < 3; ++a) {
    const float p0 = vpos[3 * (size_t)tv.x + a], p1 = vpos[3 * (size_t)tv.y + a], p2 = vpos[3 * (size_t)tv.z + a];
    l[a] = fminf(p0, fminf(p1, p2)), h[a] = fmaxf(p0, fmaxf(p1, p2));
  }
  lo[t] = make_float4(l[0], l[1], l[2], 0.f), hi[t] = make_float4(h[0], h[1], h[2], 0.f);
}

struct Box3 {
  float lx, ly, lz, hx, hy, hz;
};
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
                                               const float4* __restrict__ triHi, const float* __restrict__ skey, uint32_t leafMax, int depthCap,
                                               float4* __restrict__ scratch, uint32_t* __restrict__ cntOut,
                                               uint32_t* __restrict__ heightOut) {
  static_assert(T <= 1024 && (T & (T - 1)) == 0, "positions are 10-bit payloads of the sort keys; the bitonic network wants a power of two");
  extern __shared__ unsigned long long sub_lds[];
  // LDS carve-up (T = 1024: 8 + 8 + 40 + 4 + 24 + 4 + 8 + 12 + 8 KB = 116 KB)
  unsigned long long* key = sub_lds;                  // [T] sort keys
  unsigned long long* best = key + T;                 // [T] per segment start: (cost bits << 32 | axis << 16 | split position)
  float* eLo = reinterpret_cast<float*>(best + T);    // [3][T] element boxes, centroids (position order)
  float* eHi = eLo + 3 * T;
  float* eCen = eHi + 3 * T;                          // [4][T]: [3] = the size key (bvh_build.cpp Prim::c[3]), the fourth sweep axis
  uint32_t* eTid = reinterpret_cast<uint32_t*>(eCen + 4 * T);  // [T] triangle (reference index)
  float* sc = reinterpret_cast<float*>(eTid + T);     // [6][T] scan buffer
  float* sufA = sc + 6 * T;                           // [T] area of the suffix box
  uint16_t* rnk = reinterpret_cast<uint16_t*>(sufA + T);  // [4][T] rank of each position along each axis
  uint16_t* segS = rnk + 4 * T;                       // [T] segment [segS, segE) of the element at this position
  uint16_t* segE = segS + T;
  uint16_t* segNode = segE + T;                       // [T] local node index of the segment (all its positions carry it)
  uint16_t* segDep = segNode + T;                     // [T] depth of that node
  uint16_t* scan16 = segDep + T;                      // [T] flag scan
  uint16_t* idRank = scan16 + T;                      // [T] rank of the element's triangle id within the range: the host's tie-break
  uint16_t* ordA = idRank + T;                        // [4][T] per axis: the element (position) at each SORTED position — segment by segment
  uint16_t* nph = reinterpret_cast<uint16_t*>(sufA);  // [T] (between the sweeps and the move) new position | big << 14 | right << 15
  uint16_t* cl = nph + T;                             // [T] flag scan of the order updates
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
    eCen[3 * T + i] = skey[t];
    eTid[i] = t;
  }
  segS[i] = 0, segE[i] = (uint16_t)n, segNode[i] = 0, segDep[i] = (uint16_t)it.depth;
  if (i == 0) nodeCount = 1u, maxDep = it.depth;
  // the host builder's sorts break ties by triangle id (bvh_build.cpp split(): (key, id) is a total order): every element
  // carries the rank of its id within the range and the sort keys below hold it between the centroid and the position
  key[i] = i < n ? ((unsigned long long)order[it.b + i] << 10) | i : ~0ull;
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
  if (i < n) idRank[(uint32_t)(key[i] & 1023u)] = (uint16_t)i;
  __syncthreads();
  // The range sorted along each of the four axes by (key, id) ONCE; a split keeps every axis' order inside both children
  // (a stable partition of a sorted sequence), so the levels below re-use the orders instead of sorting again — 55 steps of
  // a bitonic network per axis and level became one flag scan.
  for (int a = 0; a < 4; ++a) {
    uint32_t ck = 0;
    if (i < n) ck = (uint32_t)fkey(eCen[a * T + i]) ^ 0x80000000u;  // order-preserving unsigned
    key[i] = i < n ? ((unsigned long long)ck << 20) | ((unsigned long long)idRank[i] << 10) | i : ~0ull;
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
    ordA[a * T + i] = i < n ? (uint16_t)(key[i] & 1023u) : (uint16_t)0;
    __syncthreads();
  }

  auto gather_boxes = [&](int a) {
    // sc[c][i] = box component c of the element that sits at sorted position i of axis a
    const uint32_t h = i < n ? (uint32_t)ordA[a * T + i] : 0u;
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
    // ---- the four sweeps (three centroid axes and the size key, as bvh_build.cpp split() below kSweepMax)
    for (int a = 0; a < 4; ++a) {
      if (i < n) rnk[a * T + (uint32_t)ordA[a * T + i]] = (uint16_t)i;
      // suffix areas: sufA[j] = area of the box of sorted positions [j, segE)
      gather_boxes(a);
      __syncthreads();
      scan_boxes(true);
      sufA[i] = area_at(i);
      __syncthreads();
      gather_boxes(a);
      __syncthreads();
      scan_boxes(false);
      // the split AFTER sorted position i: left = [segS, i + 1), right = [i + 1, segE)
      if (big && i + 1u < segE[i]) {
        const uint32_t s0 = segS[i], nl = i + 1u - s0, nr = segE[i] - i - 1u;
        const int rem = depthCap - (int)segDep[i] - 1;
        const unsigned long long maxSide = rem >= 31 ? ~0ull : (unsigned long long)leafMax << (rem < 0 ? 0 : rem);
        // (the size axis only where the segment's keys differ — sorted: its first and last key — and at 1.5 x its cost)
        const bool axisOn = a < 3 || eCen[3 * T + ordA[3 * T + segE[i] - 1u]] > eCen[3 * T + ordA[3 * T + s0]];
        if (axisOn && nl <= maxSide && nr <= maxSide) {
          float cost = area_at(i) * (float)((nl + leafMax - 1u) / leafMax) + sufA[i + 1u] * (float)((nr + leafMax - 1u) / leafMax);
          if (a == 3) cost *= 1.5f;
          if (cost == cost && cost >= 0.f && cost < inf)
            atomicMin(&best[s0], ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)a << 16) | (i + 1u));
        }
      }
      __syncthreads();
    }
    // ---- decisions: (axis, split position) per segment; no admissible split (the depth budget): the host's medianSplit —
    // the lower half along the centroid axis of the largest extent (the first of equal ones), in (key, id) order: always fits
    uint32_t ax = 0, kpos = 0;
    if (big) {
      const uint32_t s0 = segS[i], e0 = segE[i];
      const unsigned long long bb = best[s0];
      if (bb == ~0ull) {
        float ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = eCen[a * T + ordA[a * T + e0 - 1u]] - eCen[a * T + ordA[a * T + s0]];
        const uint32_t f = ext[1] > ext[0] ? 1u : 0u;
        ax = ext[2] > ext[f] ? 2u : f, kpos = s0 + (e0 - s0) / 2u;
      } else {
        ax = (uint32_t)(bb >> 16) & 7u, kpos = (uint32_t)bb & 0xffffu;
      }
    }
    // ---- every triangle to its rank along the chosen axis
    const uint32_t np = big ? rnk[ax * T + i] : i;
    // ---- the four orders follow: inside a split segment the lefts keep their order in front, the rights theirs behind (the
    // segment arrays are still the old ones here: a sorted position belongs to the same segment on every axis)
    __syncthreads();  // (sufA's last readers are done: nph / cl live there)
    nph[i] = (uint16_t)(np | (big ? 0x4000u : 0u) | ((big && np >= kpos) ? 0x8000u : 0u));
    __syncthreads();
    for (int a = 0; a < 4; ++a) {
      const uint32_t v = i < n ? (uint32_t)nph[ordA[a * T + i]] : 0u;
      const bool bigP = (v & 0x4000u) != 0u, rightP = (v & 0x8000u) != 0u;
      const uint32_t fl = (bigP && !rightP) ? 1u : 0u;
      cl[i] = (uint16_t)fl;
      __syncthreads();
      for (uint32_t d = 1; d < (uint32_t)T; d <<= 1) {
        const uint16_t add = (i < n && i >= d + segS[i]) ? cl[i - d] : (uint16_t)0;
        __syncthreads();
        cl[i] = (uint16_t)(cl[i] + add);
        __syncthreads();
      }
      uint32_t newpos = i;
      if (bigP) {
        const uint32_t s0 = segS[i], before = (uint32_t)cl[i] - fl;  // lefts of this segment in front of this position
        const unsigned long long bb = best[s0];
        const uint32_t kp = bb == ~0ull ? s0 + (uint32_t)(segE[i] - s0) / 2u : (uint32_t)bb & 0xffffu;  // (as decided above)
        newpos = rightP ? kp + (i - s0 - before) : s0 + before;
      }
      __syncthreads();
      if (i < n) ordA[a * T + newpos] = (uint16_t)(v & 0x3ffu);
      __syncthreads();
    }
    float m[10];
    uint32_t mt = 0;
    uint16_t ms = 0, me = 0, mn = 0, md = 0, mr = 0;
    if (i < n) {
      for (int c = 0; c < 3; ++c) m[c] = eLo[c * T + i], m[3 + c] = eHi[c * T + i], m[6 + c] = eCen[c * T + i];
      m[9] = eCen[3 * T + i];
      mt = eTid[i], ms = segS[i], me = segE[i], mn = segNode[i], md = segDep[i], mr = idRank[i];
    }
    __syncthreads();
    if (i < n) {
      for (int c = 0; c < 3; ++c) eLo[c * T + np] = m[c], eHi[c * T + np] = m[3 + c], eCen[c * T + np] = m[6 + c];
      eCen[3 * T + np] = m[9];
      eTid[np] = mt, idRank[np] = mr;
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

constexpr size_t subtree_lds_bytes(int T) { return (size_t)T * (8 + 8 + 40 + 4 + 24 + 4 + 8 + 12 + 8); }

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
    // (slot 0 = the left range, as the builders made it: the rotation passes run on the host builder's slot order — its candidate
    // set is not symmetric in the slots —, and k_rot_pack puts the smaller box into slot 0 afterwards)
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
  if (n < 2) return hipSuccess;
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

// ---------------------------------------------------------------- the host builder's top on the device (round 4)
// bvh_build.cpp's split() restated as kernels, rule for rule, for the ranges above the exact subtrees (more than kSubMax
// triangles): binned SAH over the three centroid axes and the size axis (64 bins from 65,536 triangles, 16 below; the size
// cut must win by 1.5 x) above 4,096 triangles, the exact sweep over the same four axes from there down, the depth budget
// (`maxSide`), strict "<" in axis-then-position order.  What the host computes with min / max and counts (centroid bounds,
// bins, boxes) does not depend on the order of the primitives, and its sorts use the total order (key, triangle id) — so the
// SAME splits come out here, and with them the host builder's tree (tests/test_gpu_bvhbuild.py compares the two), in a few
// milliseconds for a million triangles instead of the host's 35.  The host keeps what it alone can give bit for bit: the
// size keys (libm's log2f; rtbvh::planSceneExact).  A range the host would split at its MEDIAN (no admissible SAH split:
// coincident centroids, a spent depth budget) gets the host's medianSplit: in the sweep kernel and in k_subtree directly, for a
// binned range by a radix selection of the pivot and the same stable partition.
// One level of the tree per round of launches; the host keeps the list of ranges (a few thousand at most) and reads one
// split position per range and level.
constexpr uint32_t kTopChunk = 1024;   // primitives per workgroup of the per-level passes (256 threads x 4)
constexpr uint32_t kSweepMaxD = 4096;  // bvh_build.cpp kSweepMax: ranges up to this are swept exactly
constexpr int kKeyPosInf = 0x7f800000, kKeyNegInf = (int)0x807fffff;  // fkey(+inf), fkey(-inf)
struct TopItem {
  uint32_t b, e;    // range of `ord`
  uint32_t blk0;    // its first workgroup in the level's grid
  uint32_t nb;      // bins per axis (64 | 16); 0 = an exact-sweep range
  uint32_t binIdx;  // which bin block (binned ranges), which sweep slot (swept ranges)
  uint32_t node;    // its node's index
  uint32_t pad0, pad1;
};
struct TopPrep {  // per range, from its centroid bounds: bin origin and scale per axis (axis 3 = the size key)
  float lo[4], scale[4];
  uint32_t use;  // bit a: axis a has extent
  uint32_t pad[3];
};
struct TopDec {  // the chosen binned split
  int axis, bin;
  float lo, scale;
};
__device__ __forceinline__ float top_centroid(const float4& l, const float4& h, int a) {
  return a == 0 ? 0.5f * l.x + 0.5f * h.x : a == 1 ? 0.5f * l.y + 0.5f * h.y : 0.5f * l.z + 0.5f * h.z;
}
__device__ __forceinline__ int top_bin(float c, float lo, float scale, int NB) {
  const int k = (int)((c - lo) * scale);
  return min(NB - 1, max(0, k));
}
// block-wide min / max of `cnt` ordered-int values per thread; the result is valid in thread 0
template <int CNT>
__device__ __forceinline__ void block_minmax(int* mn, int* mx, int* sh /* [2 * CNT * 4] */) {
  for (int c = 0; c < CNT; ++c)
    for (int off = 32; off > 0; off >>= 1) {
      mn[c] = min(mn[c], __shfl_xor(mn[c], off, 64));
      mx[c] = max(mx[c], __shfl_xor(mx[c], off, 64));
    }
  const uint32_t w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0)
    for (int c = 0; c < CNT; ++c) sh[(2 * c) * 4 + w] = mn[c], sh[(2 * c + 1) * 4 + w] = mx[c];
  __syncthreads();
  if (threadIdx.x == 0)
    for (int c = 0; c < CNT; ++c)
      for (int k = 0; k < 4; ++k) mn[c] = min(mn[c], sh[(2 * c) * 4 + k]), mx[c] = max(mx[c], sh[(2 * c + 1) * 4 + k]);
}

__global__ void k_top_init(uint32_t count, int* __restrict__ ib, int* __restrict__ cbx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count * 8u) ib[i] = (i & 7u) < 4u ? kKeyPosInf : kKeyNegInf;
  if (i < count * 12u) cbx[i] = (i % 6u) < 3u ? kKeyPosInf : kKeyNegInf;
}
// centroid bounds and size-key range of every range (bvh_build.cpp split(): cb, sLo, sHi)
__global__ __launch_bounds__(256) void k_top_bounds(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                    const uint32_t* __restrict__ ord, const float4* __restrict__ lo,
                                                    const float4* __restrict__ hi, const float* __restrict__ skey, int* __restrict__ ib) {
  __shared__ int sh[32];
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk;
  int mn[4] = {kKeyPosInf, kKeyPosInf, kKeyPosInf, kKeyPosInf}, mx[4] = {kKeyNegInf, kKeyNegInf, kKeyNegInf, kKeyNegInf};
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + threadIdx.x + 256u * e;
    if (p < it.e) {
      const uint32_t id = ord[p];
      const float4 l = lo[id], h = hi[id];
      const float v[4] = {top_centroid(l, h, 0), top_centroid(l, h, 1), top_centroid(l, h, 2), skey[id]};
      for (int c = 0; c < 4; ++c) mn[c] = min(mn[c], fkey(v[c])), mx[c] = max(mx[c], fkey(v[c]));
    }
  }
  block_minmax<4>(mn, mx, sh);
  if (threadIdx.x == 0)
    for (int c = 0; c < 4; ++c) atomicMin(&ib[8 * r + c], mn[c]), atomicMax(&ib[8 * r + 4 + c], mx[c]);
}
__global__ void k_top_prep(const TopItem* __restrict__ items, uint32_t count, const int* __restrict__ ib, TopPrep* __restrict__ prep,
                           int* __restrict__ bins) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  const TopItem it = items[r];
  TopPrep P;
  P.use = 0, P.pad[0] = P.pad[1] = P.pad[2] = 0;
  const int NB = it.nb ? (int)it.nb : 1;
  for (int a = 0; a < 4; ++a) {
    const float l = funkey(ib[8 * r + a]), h = funkey(ib[8 * r + 4 + a]);
    const float ext = h - l;
    const bool use = ext > 0.f;
    P.lo[a] = l, P.scale[a] = use ? NB / ext : 0.f;
    P.use |= use ? 1u << a : 0u;
  }
  prep[r] = P;
  if (it.nb) {  // its bins: empty boxes, no primitives
    int* B = bins + (size_t)it.binIdx * 4u * 64u * 7u;
    for (uint32_t k = 0; k < 4u * 64u; ++k) {
      for (int c = 0; c < 3; ++c) B[7 * k + c] = kKeyPosInf, B[7 * k + 3 + c] = kKeyNegInf;
      B[7 * k + 6] = 0;
    }
  }
}
// the bins of all four axes in one pass (LDS bins per workgroup, then merged: min / max and counts)
__global__ __launch_bounds__(256) void k_top_bins(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                  const uint32_t* __restrict__ ord, const float4* __restrict__ lo,
                                                  const float4* __restrict__ hi, const float* __restrict__ skey,
                                                  const TopPrep* __restrict__ prep, int* __restrict__ bins) {
  __shared__ int sb[4 * 64 * 7];
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  if (it.nb == 0) return;
  const int NB = (int)it.nb;
  const TopPrep P = prep[r];
  for (uint32_t k = threadIdx.x; k < 4u * 64u; k += 256u) {
    for (int c = 0; c < 3; ++c) sb[7 * k + c] = kKeyPosInf, sb[7 * k + 3 + c] = kKeyNegInf;
    sb[7 * k + 6] = 0;
  }
  __syncthreads();
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk;
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + threadIdx.x + 256u * e;
    if (p < it.e) {
      const uint32_t id = ord[p];
      const float4 l = lo[id], h = hi[id];
      const float v[4] = {top_centroid(l, h, 0), top_centroid(l, h, 1), top_centroid(l, h, 2), skey[id]};
      const int bl[3] = {fkey(l.x), fkey(l.y), fkey(l.z)}, bh[3] = {fkey(h.x), fkey(h.y), fkey(h.z)};
      for (int a = 0; a < 4; ++a) {
        if (!((P.use >> a) & 1u)) continue;
        int* B = sb + 7 * (a * 64 + top_bin(v[a], P.lo[a], P.scale[a], NB));
        for (int c = 0; c < 3; ++c) atomicMin(&B[c], bl[c]), atomicMax(&B[3 + c], bh[c]);
        atomicAdd(&B[6], 1);
      }
    }
  }
  __syncthreads();
  int* G = bins + (size_t)it.binIdx * 4u * 64u * 7u;
  for (uint32_t k = threadIdx.x; k < 4u * 64u; k += 256u) {
    const int cnt = sb[7 * k + 6];
    if (cnt) {
      for (int c = 0; c < 3; ++c) atomicMin(&G[7 * k + c], sb[7 * k + c]), atomicMax(&G[7 * k + 3 + c], sb[7 * k + 3 + c]);
      atomicAdd(&G[7 * k + 6], cnt);
    }
  }
}
__device__ __forceinline__ Box3 top_box_of(const int* B) {
  return Box3{funkey(B[0]), funkey(B[1]), funkey(B[2]), funkey(B[3]), funkey(B[4]), funkey(B[5])};
}
// the binned SAH choice of bvh_build.cpp split(): one WAVE per range, lane = bin.  The host walks the bins sequentially — a
// suffix pass for the right boxes and counts, a prefix pass pricing the cut behind every bin, the first strict minimum in
// axis-then-bin order —; box unions are min / max and the counts integers, so the scans give the same boxes, the same areas,
// the same costs, and the wave picks the minimum of (cost, axis, bin) — the sequential loop's choice.
__global__ __launch_bounds__(64) void k_top_choose(const TopItem* __restrict__ items, uint32_t count, uint32_t depth, int depthCap,
                                                   uint32_t leafMax, const TopPrep* __restrict__ prep, const int* __restrict__ bins,
                                                   TopDec* __restrict__ dec, uint32_t* __restrict__ mid, uint32_t* __restrict__ needMedian) {
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  if (r >= count) return;
  const TopItem it = items[r];
  if (it.nb == 0) return;
  const int NB = (int)it.nb;
  const uint32_t n = it.e - it.b;
  const TopPrep P = prep[r];
  const int rem = depthCap - (int)depth - 1;
  const unsigned long long maxSide = rem >= 31 ? ~0ull : (unsigned long long)leafMax << (rem < 0 ? 0 : rem);
  const int* G = bins + (size_t)it.binIdx * 4u * 64u * 7u;
  const float inf = __int_as_float(0x7f800000);
  unsigned long long best = ~0ull;  // cost bits << 32 | axis << 24 | bin << 16 ... (the left count rides separately)
  uint32_t bestLeft = 0;
  for (int ax = 0; ax < 4; ++ax) {
    if (!((P.use >> ax) & 1u)) continue;  // (wave-uniform)
    const bool in = (int)lane < NB;
    const int* A = G + 7 * (ax * 64 + (in ? (int)lane : 0));
    Box3 mine = in ? top_box_of(A) : Box3{inf, inf, inf, -inf, -inf, -inf};
    uint32_t cnt = in ? (uint32_t)A[6] : 0u;
    // inclusive prefix (bins 0 .. lane) and inclusive suffix (bins lane .. NB - 1) of boxes and counts
    Box3 pre = mine, suf = mine;
    uint32_t cpre = cnt, csuf = cnt;
    for (int off = 1; off < 64; off <<= 1) {
      const Box3 pb{__shfl_up(pre.lx, off, 64), __shfl_up(pre.ly, off, 64), __shfl_up(pre.lz, off, 64),
                    __shfl_up(pre.hx, off, 64), __shfl_up(pre.hy, off, 64), __shfl_up(pre.hz, off, 64)};
      const uint32_t pc = __shfl_up(cpre, off, 64);
      if ((int)lane >= off) pre = rot_union(pre, pb), cpre += pc;
      const Box3 sb{__shfl_down(suf.lx, off, 64), __shfl_down(suf.ly, off, 64), __shfl_down(suf.lz, off, 64),
                    __shfl_down(suf.hx, off, 64), __shfl_down(suf.hy, off, 64), __shfl_down(suf.hz, off, 64)};
      const uint32_t sc = __shfl_down(csuf, off, 64);
      if ((int)lane + off < 64) suf = rot_union(suf, sb), csuf += sc;
    }
    // the cut behind bin k = lane: left = bins 0 .. k, right = bins k + 1 .. NB - 1 (lane k + 1's suffix)
    const float rArea = __shfl_down(half_area(suf), 1, 64);
    const uint32_t rCnt = __shfl_down(csuf, 1, 64);
    unsigned long long key = ~0ull;
    if ((int)lane < NB - 1 && cpre != 0u && rCnt != 0u && cpre <= maxSide && rCnt <= maxSide) {
      float cost = half_area(pre) * ceilf(cpre / (float)leafMax) + rArea * ceilf(rCnt / (float)leafMax);
      if (ax == 3) cost *= 1.5f;  // (bvh_build.cpp sizeBias)
      if (cost == cost && cost >= 0.f && cost < inf) key = ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)ax << 8) | lane;
    }
    unsigned long long m = key;
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off, 64);
      m = o < m ? o : m;
    }
    if (m < best) {  // (strictly cheaper than every earlier axis' best, or the first)
      best = m;
      bestLeft = __shfl(cpre, (int)(m & 63u), 64);
    }
  }
  if (lane != 0) return;
  // the host's median fall-backs: no admissible cut, or an extremely lopsided one deep in the tree
  const uint32_t small = bestLeft < n - bestLeft ? bestLeft : n - bestLeft;
  if (best == ~0ull || (n > 64u && (unsigned long long)small * 64ull < n && (int)depth > rtbvh::kMaxDepth / 2)) {
    atomicExch(needMedian, 1u);  // (a median split: the partition passes leave the range alone, the host selects its pivot)
    dec[r] = TopDec{-1, 0, 0.f, 0.f}, mid[r] = it.b + n / 2u;
    return;
  }
  const int bestAxis = (int)((best >> 8) & 3u), bestBin = (int)(best & 63u);
  dec[r] = TopDec{bestAxis, bestBin, P.lo[bestAxis], P.scale[bestAxis]};
  mid[r] = it.b + bestLeft;
}
// (key, id) of the host's medianSplit order as one 64-bit word
__device__ __forceinline__ unsigned long long top_median_key(uint32_t id, int axis, const float4* lo, const float4* hi) {
  return ((unsigned long long)((uint32_t)fkey(top_centroid(lo[id], hi[id], axis)) ^ 0x80000000u) << 32) | id;
}
__device__ __forceinline__ bool top_goes_left(const TopDec& D, int NB, uint32_t id, const float4* lo, const float4* hi, const float* skey) {
  if (D.axis >= 4) {  // a MEDIAN split along axis D.axis - 4: everything below the pivot key {lo, scale} = its two halves' bits
    const unsigned long long pivot = ((unsigned long long)__float_as_uint(D.lo) << 32) | __float_as_uint(D.scale);
    return top_median_key(id, D.axis - 4, lo, hi) < pivot;
  }
  const float v = D.axis == 3 ? skey[id] : top_centroid(lo[id], hi[id], D.axis);
  return top_bin(v, D.lo, D.scale, NB) <= D.bin;
}
// radix selection of a range's k-th smallest median key, one byte per pass: the histogram of the byte at `shift` among the keys
// that agree with `prefix` above it
__global__ __launch_bounds__(256) void k_top_select_hist(const uint32_t* __restrict__ ord, uint32_t b, uint32_t e, int axis,
                                                         const float4* __restrict__ lo, const float4* __restrict__ hi,
                                                         unsigned long long prefix, int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t p = b + blockIdx.x * 256u + threadIdx.x; p < e; p += gridDim.x * 256u) {
    const unsigned long long k = top_median_key(ord[p], axis, lo, hi);
    if (shift >= 56 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&sh[(uint32_t)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}
// stable partition of the binned ranges, three passes: lefts per workgroup, their prefix per range, the move
__global__ __launch_bounds__(256) void k_top_count(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                   const uint32_t* __restrict__ ord, const float4* __restrict__ lo,
                                                   const float4* __restrict__ hi, const float* __restrict__ skey,
                                                   const TopDec* __restrict__ dec, uint32_t* __restrict__ blockLeft) {
  __shared__ uint32_t sh[4];
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  if (it.nb == 0) return;
  const TopDec D = dec[r];
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk;
  uint32_t c = 0;
  if (D.axis >= 0)
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t p = p0 + threadIdx.x + 256u * e;
      if (p < it.e) c += top_goes_left(D, (int)it.nb, ord[p], lo, hi, skey) ? 1u : 0u;
    }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockLeft[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// one wave per range: exclusive prefix of its workgroups' counts (in place)
__global__ __launch_bounds__(64) void k_top_offsets(const TopItem* __restrict__ items, uint32_t* __restrict__ blockLeft) {
  const TopItem it = items[blockIdx.x];
  if (it.nb == 0) return;
  const uint32_t nBlk = (it.e - it.b + kTopChunk - 1u) / kTopChunk, lane = threadIdx.x;
  uint32_t base = 0;
  for (uint32_t k0 = 0; k0 < nBlk; k0 += 64u) {
    const uint32_t k = k0 + lane;
    const uint32_t v = k < nBlk ? blockLeft[it.blk0 + k] : 0u;
    uint32_t s = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(s, off, 64);
      if ((int)lane >= off) s += o;
    }
    if (k < nBlk) blockLeft[it.blk0 + k] = base + s - v;
    base += __shfl(s, 63, 64);
  }
}
__global__ __launch_bounds__(256) void k_top_scatter(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                     const uint32_t* __restrict__ ord, const float4* __restrict__ lo,
                                                     const float4* __restrict__ hi, const float* __restrict__ skey,
                                                     const TopDec* __restrict__ dec, const uint32_t* __restrict__ mid,
                                                     const uint32_t* __restrict__ blockLeft, uint32_t* __restrict__ tmp) {
  __shared__ uint32_t sh[4];
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  if (it.nb == 0) return;
  const TopDec D = dec[r];
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk;
  // thread t owns the four consecutive positions p0 + 4 t ... (the partition is stable)
  uint32_t id[4];
  bool left[4];
  uint32_t c = 0;
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + 4u * threadIdx.x + e;
    id[e] = p < it.e ? ord[p] : 0u;
    left[e] = p < it.e && D.axis >= 0 && top_goes_left(D, (int)it.nb, id[e], lo, hi, skey);
    c += left[e] ? 1u : 0u;
  }
  uint32_t s = c;  // inclusive scan over the workgroup's threads
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(s, off, 64);
    if ((int)lane >= off) s += o;
  }
  if (lane == 63u) sh[w] = s;
  __syncthreads();
  uint32_t before = s - c;
  for (uint32_t k = 0; k < w; ++k) before += sh[k];
  if (D.axis < 0) {  // (no decision: the range stays as it is)
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t p = p0 + 4u * threadIdx.x + e;
      if (p < it.e) tmp[p] = id[e];
    }
    return;
  }
  const uint32_t leftBase = it.b + blockLeft[blockIdx.x];
  // rights before this workgroup's chunk = elements before it minus lefts before it
  const uint32_t rightBase = mid[r] + (p0 - it.b) - blockLeft[blockIdx.x];
  uint32_t l = before, rgt = 4u * threadIdx.x - before;
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + 4u * threadIdx.x + e;
    if (p >= it.e) break;
    if (left[e]) tmp[leftBase + l++] = id[e];
    else tmp[rightBase + rgt++] = id[e];
  }
}
__global__ __launch_bounds__(256) void k_top_copyback(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                      const uint32_t* __restrict__ tmp, uint32_t* __restrict__ ord) {
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  if (it.nb == 0) return;
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk;
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + threadIdx.x + 256u * e;
    if (p < it.e) ord[p] = tmp[p];
  }
}

// The exact sweep of bvh_build.cpp split() for one range of 1,025 ... 4,096 primitives per workgroup of 1,024 threads (four
// elements each): the range in ascending triangle id (so that a position stands for the id in the sort keys), then per axis a
// bitonic sort by (key, id), suffix areas, prefix boxes, the cost of every cut, the cheapest — lowest axis, then lowest
// position, among equal costs — and the range rewritten in the best axis' order.
constexpr uint32_t kSweepT = 1024, kSweepN = 4096;
constexpr size_t kSweepLds = (size_t)kSweepN * (8 + 24 + 4 + 2) + 64;
__device__ __forceinline__ void sweep_sort(unsigned long long* key) {
  for (uint32_t k = 2; k <= kSweepN; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t e = 0; e < 2u; ++e) {
        const uint32_t p = threadIdx.x + kSweepT * e;          // pair index 0 .. 2047
        const uint32_t i = ((p & ~(j - 1u)) << 1) | (p & (j - 1u)), x = i | j;
        const unsigned long long ka = key[i], kb = key[x];
        const bool up = (i & k) == 0;
        if ((ka > kb) == up) key[i] = kb, key[x] = ka;
      }
      __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_top_sweep(const TopItem* __restrict__ items, const uint32_t* __restrict__ sweepList,
                                                    uint32_t depth, int depthCap, uint32_t leafMax, const int* __restrict__ ib,
                                                    const float4* __restrict__ lo, const float4* __restrict__ hi,
                                                    const float* __restrict__ skey, uint32_t* __restrict__ ord, uint32_t* __restrict__ tmp,
                                                    uint32_t* __restrict__ mid) {
  extern __shared__ unsigned long long sweep_lds[];
  unsigned long long* key = sweep_lds;                          // [4096] sort keys
  float* sc = reinterpret_cast<float*>(key + kSweepN);          // [6][4096] boxes in sorted order / their scans
  float* sufA = sc + 6 * kSweepN;                               // [4096] area of the sorted suffix [i, n)
  uint16_t* bestPos = reinterpret_cast<uint16_t*>(sufA + kSweepN);  // [4096] home positions in the best axis' order
  unsigned long long* best = reinterpret_cast<unsigned long long*>(bestPos + kSweepN);  // cost bits << 32 | axis << 16 | left count
  const uint32_t r = sweepList[blockIdx.x];
  const TopItem it = items[r];
  const uint32_t n = it.e - it.b, t = threadIdx.x;
  const float inf = __int_as_float(0x7f800000);
  // home order = ascending triangle id
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t j = t + kSweepT * e;
    key[j] = j < n ? (unsigned long long)ord[it.b + j] : ~0ull;
  }
  if (t == 0) *best = ~0ull;
  __syncthreads();
  sweep_sort(key);
  uint32_t* home = tmp + it.b;  // (the range's slice of the second buffer: nobody else's)
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t j = t + kSweepT * e;
    if (j < n) home[j] = (uint32_t)key[j];
  }
  __syncthreads();
  const int rem = depthCap - (int)depth - 1;
  const unsigned long long maxSide = rem >= 31 ? ~0ull : (unsigned long long)leafMax << (rem < 0 ? 0 : rem);
  const float sLo = funkey(ib[8 * r + 3]), sHi = funkey(ib[8 * r + 7]);
  auto gather = [&]() {
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t i = t + kSweepT * e;
      if (i < n) {
        const uint32_t id = home[(uint32_t)(key[i] & 4095u)];
        const float4 l = lo[id], h = hi[id];
        sc[i] = l.x, sc[kSweepN + i] = l.y, sc[2 * kSweepN + i] = l.z, sc[3 * kSweepN + i] = h.x, sc[4 * kSweepN + i] = h.y, sc[5 * kSweepN + i] = h.z;
      } else {
        sc[i] = sc[kSweepN + i] = sc[2 * kSweepN + i] = inf, sc[3 * kSweepN + i] = sc[4 * kSweepN + i] = sc[5 * kSweepN + i] = -inf;
      }
    }
    __syncthreads();
  };
  auto scan = [&](bool suffix) {  // inclusive scan (union of boxes) over the sorted positions
    for (uint32_t d = 1; d < kSweepN; d <<= 1) {
      float v[4][6];
      for (uint32_t e = 0; e < 4u; ++e) {
        const uint32_t i = t + kSweepT * e;
        const bool take = suffix ? i + d < kSweepN : i >= d;
        const uint32_t j = suffix ? i + d : i - d;
        for (int c = 0; c < 6; ++c) v[e][c] = sc[c * kSweepN + i];
        if (take)
          for (int c = 0; c < 3; ++c) v[e][c] = fminf(v[e][c], sc[c * kSweepN + j]), v[e][3 + c] = fmaxf(v[e][3 + c], sc[(3 + c) * kSweepN + j]);
      }
      __syncthreads();
      for (uint32_t e = 0; e < 4u; ++e) {
        const uint32_t i = t + kSweepT * e;
        for (int c = 0; c < 6; ++c) sc[c * kSweepN + i] = v[e][c];
      }
      __syncthreads();
    }
  };
  auto area_at = [&](uint32_t j) {
    const float dx = sc[3 * kSweepN + j] - sc[j], dy = sc[4 * kSweepN + j] - sc[kSweepN + j], dz = sc[5 * kSweepN + j] - sc[2 * kSweepN + j];
    return dx < 0.f ? 0.f : dx * dy + dy * dz + dz * dx;
  };
  for (int ax = 0; ax < 4; ++ax) {
    if (ax == 3 && !(sHi > sLo)) continue;  // (wave-uniform: every thread reads the same bounds)
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t j = t + kSweepT * e;
      if (j < n) {
        const uint32_t id = home[j];
        const float v = ax == 3 ? skey[id] : top_centroid(lo[id], hi[id], ax);
        key[j] = ((unsigned long long)((uint32_t)fkey(v) ^ 0x80000000u) << 12) | j;
      } else {
        key[j] = ~0ull;
      }
    }
    __syncthreads();
    sweep_sort(key);
    gather();
    scan(true);
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t i = t + kSweepT * e;
      sufA[i] = area_at(i);
    }
    __syncthreads();
    gather();
    scan(false);
    const unsigned long long before = *best;
    __syncthreads();
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t i = t + kSweepT * e;  // the cut behind sorted position i: i + 1 elements on the left
      if (i + 1u < n) {
        const uint32_t nl = i + 1u, nr = n - nl;
        if (nl <= maxSide && nr <= maxSide) {
          float cost = area_at(i) * ceilf(nl / (float)leafMax) + sufA[i + 1u] * ceilf(nr / (float)leafMax);
          if (ax == 3) cost *= 1.5f;
          if (cost == cost && cost >= 0.f && cost < inf)
            atomicMin(best, ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)ax << 16) | nl);
        }
      }
    }
    __syncthreads();
    if (*best != before)  // this axis holds the best cut so far: keep its order
      for (uint32_t e = 0; e < 4u; ++e) {
        const uint32_t i = t + kSweepT * e;
        bestPos[i] = (uint16_t)(key[i] & 4095u);
      }
    __syncthreads();
  }
  const unsigned long long bb = *best;
  if (bb == ~0ull) {
    // no admissible cut (the depth budget): the host's medianSplit — the lower half along the centroid axis of the largest
    // extent (the first of equal ones: std::sort's insertion sort of three), in (key, id) order
    float ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = funkey(ib[8 * r + 4 + a]) - funkey(ib[8 * r + a]);
    const int f = ext[1] > ext[0] ? 1 : 0, ax = ext[2] > ext[f] ? 2 : f;
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t j = t + kSweepT * e;
      if (j < n) {
        const uint32_t id = home[j];
        key[j] = ((unsigned long long)((uint32_t)fkey(top_centroid(lo[id], hi[id], ax)) ^ 0x80000000u) << 12) | j;
      } else {
        key[j] = ~0ull;
      }
    }
    __syncthreads();
    sweep_sort(key);
    for (uint32_t e = 0; e < 4u; ++e) {
      const uint32_t i = t + kSweepT * e;
      if (i < n) ord[it.b + i] = home[(uint32_t)(key[i] & 4095u)];
    }
    if (t == 0) mid[r] = it.b + n / 2u;
    return;
  }
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t i = t + kSweepT * e;
    if (i < n) ord[it.b + i] = home[bestPos[i]];
  }
  if (t == 0) mid[r] = it.b + (uint32_t)(bb & 0xffffu);
}

// boxes of the two children of every split range
__global__ __launch_bounds__(256) void k_top_childbox(const TopItem* __restrict__ items, const uint32_t* __restrict__ blkItem,
                                                      const uint32_t* __restrict__ ord, const float4* __restrict__ lo,
                                                      const float4* __restrict__ hi, const uint32_t* __restrict__ mid, int* __restrict__ cbx) {
  __shared__ int sh[48];
  const uint32_t r = blkItem[blockIdx.x];
  const TopItem it = items[r];
  const uint32_t p0 = it.b + (blockIdx.x - it.blk0) * kTopChunk, m = mid[r];
  int mn[6], mx[6];  // [side * 3 + axis]
  for (int c = 0; c < 6; ++c) mn[c] = kKeyPosInf, mx[c] = kKeyNegInf;
  for (uint32_t e = 0; e < 4u; ++e) {
    const uint32_t p = p0 + threadIdx.x + 256u * e;
    if (p < it.e) {
      const uint32_t id = ord[p];
      const float4 l = lo[id], h = hi[id];
      const int s = p >= m ? 3 : 0;
      mn[s] = min(mn[s], fkey(l.x)), mn[s + 1] = min(mn[s + 1], fkey(l.y)), mn[s + 2] = min(mn[s + 2], fkey(l.z));
      mx[s] = max(mx[s], fkey(h.x)), mx[s + 1] = max(mx[s + 1], fkey(h.y)), mx[s + 2] = max(mx[s + 2], fkey(h.z));
    }
  }
  block_minmax<6>(mn, mx, sh);
  if (threadIdx.x == 0)
    for (int s = 0; s < 2; ++s)
      for (int a = 0; a < 3; ++a) atomicMin(&cbx[12 * r + 6 * s + a], mn[3 * s + a]), atomicMax(&cbx[12 * r + 6 * s + 3 + a], mx[3 * s + a]);
}
// the node record of every split range (padded child boxes, the left range in slot 0), its parts registered for k_subtree, its
// leaves put in ascending id
__global__ void k_top_emit(const TopItem* __restrict__ items, uint32_t count, uint32_t depth, const uint32_t* __restrict__ mid,
                           const int32_t* __restrict__ refs, const int* __restrict__ cbx, float pad, uint32_t* __restrict__ ord,
                           RNode* __restrict__ nodes, SubItem* __restrict__ subs) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  const TopItem it = items[r];
  const uint32_t m = mid[r];
  Box3 B[2];
  for (int s = 0; s < 2; ++s) {
    B[s] = top_box_of(cbx + 12 * r + 6 * s);
    B[s].lx -= pad, B[s].ly -= pad, B[s].lz -= pad, B[s].hx += pad, B[s].hy += pad, B[s].hz += pad;
  }
  RNode N;
  N.pad[0] = N.pad[1] = 0;
  for (int s = 0; s < 2; ++s) {
    const int slot = s;  // (the left range in slot 0: the rotation passes want the host builder's slot order)
    const uint32_t cb = s ? m : it.b, ce = s ? it.e : m;
    int32_t ref = refs[2 * r + s];
    if (ref < 0 && ((~(uint32_t)ref) & rtbvh::kPartFlag)) {
      subs[(~(uint32_t)ref) & (rtbvh::kPartFlag - 1u)] = SubItem{cb, ce, depth + 1u, it.node, (uint32_t)slot};
      ref = 0;  // (patched by k_sub_relocate once the subtree's root has its index)
    } else if (ref < 0) {
      for (uint32_t x = cb + 1u; x < ce; ++x) {  // a leaf of the top: ascending triangle id, as the host leaves it
        const uint32_t v = ord[x];
        uint32_t y = x;
        while (y > cb && ord[y - 1u] > v) ord[y] = ord[y - 1u], --y;
        ord[y] = v;
      }
    }
    rot_set(N, slot, ref, B[s]);
  }
  nodes[it.node] = N;
}
__global__ void k_iota(uint32_t* __restrict__ v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// dynamic LDS beyond 64 KiB is granted per kernel AND per device (one process may build on several devices: rt_group)
static hipError_t grant_lds(const void* fn, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = -1;
  if (dev >= 0 && (done.load() & (1ull << dev))) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && dev >= 0) done.fetch_or(1ull << dev);
  return e;
}

// One allocation cut into the builder's temporaries (two dozen hipMalloc / hipFree pairs cost 2-3 ms of a 20-ms build).
struct Arena {
  char* base = nullptr;
  size_t size = 0, used = 0;
  size_t reserve(size_t bytes) {
    const size_t at = size;
    size += (bytes + 255u) & ~(size_t)255u;
    return at;
  }
  hipError_t commit() { return hipMalloc((void**)&base, size ? size : 256); }
  template <class T>
  T* at(size_t off) const { return reinterpret_cast<T*>(base + off); }
  void release() {
    if (base) (void)hipFree(base);
    base = nullptr;
  }
};

// The part of a build below a finished top: nodesF[0, nTop) hold the top's float records (part refs still 0), nodes16 the
// same packed, `subs` the parts; builds every part's exact subtree, numbers them behind the top, runs the rotation passes
// and the final numbering / packing, and writes the triangle records.  Outputs as gpu_bvh_build.
static hipError_t subtrees_and_finish(const float* dVpos, const uint4* dTriShade, uint32_t n, uint32_t nTop, uint32_t nSub,
                                      const SubItem* subs, uint32_t* order, const float4* lo, const float4* hi, const float* skey, uint32_t leafMax,
                                      int depthCap, float pad, float boxScale, uint32_t topMaxDepth, uint32_t maxNodes, float4* nodesF,
                                      uint4* nodes16, float4* tris, float4* trisRef, uint32_t* nTotalOut, uint32_t* maxDepthOut,
                                      hipStream_t stream) {
  Arena A;
  const size_t nS = nSub ? nSub : 1u;
  size_t scanBytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, scanBytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, nS, rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  const size_t oSizes = A.reserve(nS * 4), oScratchOff = A.reserve(nS * 4), oSubNodes = A.reserve(nS * 4), oFinalOff = A.reserve(nS * 4),
               oHeight = A.reserve(4), oTmp = A.reserve(scanBytes ? scanBytes : 16), oScratch = A.reserve(4 * (size_t)n * sizeof(float4));
  if ((e = A.commit()) != hipSuccess) return e;
  uint32_t *subSizes = A.at<uint32_t>(oSizes), *scratchOff = A.at<uint32_t>(oScratchOff), *subNodes = A.at<uint32_t>(oSubNodes),
           *finalOff = A.at<uint32_t>(oFinalOff), *height = A.at<uint32_t>(oHeight);
  void* tmp = A.at<void>(oTmp);
  float4* scratch = A.at<float4>(oScratch);
#define SF_TRY(expr)        \
  do {                      \
    e = (expr);             \
    if (e != hipSuccess) {  \
      A.release();          \
      return e;             \
    }                       \
  } while (0)
  const dim3 blk(256), grdN((n + 255) / 256);
  uint32_t nTotal = nTop, maxDepth = topMaxDepth;
  SF_TRY(hipMemsetAsync(height, 0, sizeof(uint32_t), stream));
  if (nSub) {
    hipLaunchKernelGGL(k_sub_sizes, dim3((nSub + 255) / 256), blk, 0, stream, subs, nSub, subSizes);
    SF_TRY(rocprim::exclusive_scan(tmp, scanBytes, subSizes, scratchOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    const size_t ldsBytes = subtree_lds_bytes((int)kSubMax);
    static std::atomic<unsigned long long> ldsSet{0};
    SF_TRY(grant_lds(reinterpret_cast<const void*>(&k_subtree<(int)kSubMax>), ldsBytes, ldsSet));
    hipLaunchKernelGGL((k_subtree<(int)kSubMax>), dim3(nSub), dim3(kSubMax), ldsBytes, stream, subs, scratchOff, order, lo, hi, skey, leafMax,
                       depthCap, scratch, subNodes, height);
    SF_TRY(rocprim::exclusive_scan(tmp, scanBytes, subNodes, finalOff, 0u, (size_t)nSub, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_sub_relocate, dim3(nSub), blk, 0, stream, subs, scratchOff, subNodes, finalOff, nTop, scratch, pad, boxScale,
                       nodes16, nodesF);
    uint32_t lastOff = 0, lastCnt = 0, h = 0;
    SF_TRY(hipMemcpyAsync(&lastOff, finalOff + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    SF_TRY(hipMemcpyAsync(&lastCnt, subNodes + (nSub - 1), 4, hipMemcpyDeviceToHost, stream));
    SF_TRY(hipMemcpyAsync(&h, height, 4, hipMemcpyDeviceToHost, stream));
    SF_TRY(hipStreamSynchronize(stream));
    nTotal = nTop + lastOff + lastCnt;
    maxDepth = maxDepth > h ? maxDepth : h;
    if (nTotal > maxNodes) {
      A.release();
      return hipErrorInvalidValue;
    }
  }
  // rotation passes over the whole tree (as the host builder's: 3 on big scenes, 8 on small ones)
  SF_TRY(rotate_and_pack(nodesF, nodes16, nTotal, depthCap, boxScale, n > 200000u ? 3 : 8, &maxDepth, stream));
  // triangle records in the FINAL leaf order (the exact builder has reordered its ranges), and in reference order
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, order, n, tris);
  hipLaunchKernelGGL(k_tri_records, grdN, blk, 0, stream, dVpos, dTriShade, (const uint32_t*)nullptr, n, trisRef);
  SF_TRY(hipStreamSynchronize(stream));
  SF_TRY(hipGetLastError());
#undef SF_TRY
  A.release();
  *nTotalOut = nTotal, *maxDepthOut = maxDepth;
  return hipSuccess;
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

// The HYBRID build (rt_options.bvh_builder = RT_BVH_HYBRID): the host builder's own top (rtbvh::buildTop: its split
// choices down to parts of <= kSubMax triangles) and, below it, the exact subtrees of step 6 — one workgroup per part.
// The host spends most of a build in those bottom levels (sorts of every range of <= 4,096 triangles along four axes);
// the top is a few binned passes.  Same arrays out as gpu_bvh_build.
hipError_t gpu_bvh_build_over_top(const float* dVpos, const uint4* dTriShade, const float* hSizeKey, uint32_t n, const rtbvh::TopBuilt& top,
                                  GpuBvh* out, hipStream_t stream) {
  *out = GpuBvh{};
  if (!hSizeKey) return hipErrorInvalidValue;
  const uint32_t nTop = (uint32_t)top.nodes.size(), nSub = (uint32_t)top.parts.size();
  if (n == 0 || nTop == 0 || top.order.size() != n) return hipErrorInvalidValue;
  for (const rtbvh::TopBuilt::Part& p : top.parts)
    if (p.e <= p.b || p.e - p.b > kSubMax || p.e > n || p.parent >= nTop || p.slot > 1u) return hipErrorInvalidValue;
  const uint32_t maxNodes = n + nTop;
  const size_t nS = nSub ? nSub : 1u;
  Arena A;
  const size_t oLo = A.reserve((size_t)n * sizeof(float4)), oHi = A.reserve((size_t)n * sizeof(float4)), oOrd = A.reserve((size_t)n * 4),
               oSubs = A.reserve(nS * sizeof(SubItem)), oKey = A.reserve((size_t)n * 4);
  float4 *nodesF = nullptr, *tris = nullptr, *trisRef = nullptr;
  uint4* nodes16 = nullptr;
  bool keepOutputs = false;
  auto cleanup = [&]() {
    A.release();
    if (!keepOutputs)
      for (void* p : {(void*)nodes16, (void*)nodesF, (void*)tris, (void*)trisRef})
        if (p) (void)hipFree(p);
  };
  GB_TRY(A.commit());
  GB_TRY(hipMalloc((void**)&nodes16, 2 * (size_t)maxNodes * sizeof(uint4)));
  GB_TRY(hipMalloc((void**)&nodesF, 4 * (size_t)maxNodes * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&tris, 3 * (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&trisRef, 3 * (size_t)n * sizeof(float4)));
  float4 *lo = A.at<float4>(oLo), *hi = A.at<float4>(oHi);
  uint32_t* order = A.at<uint32_t>(oOrd);
  SubItem* subs = A.at<SubItem>(oSubs);
  float* skey = A.at<float>(oKey);
  GB_TRY(hipMemcpyAsync(order, top.order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  GB_TRY(hipMemcpyAsync(skey, hSizeKey, (size_t)n * 4, hipMemcpyHostToDevice, stream));
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
  hipLaunchKernelGGL(k_tri_boxes, dim3((n + 255) / 256), dim3(256), 0, stream, dVpos, dTriShade, n, lo, hi);
  GB_TRY(hipStreamSynchronize(stream));  // (the host vectors above go out of scope)
  uint32_t nTotal = 0, maxDepth = 0;
  GB_TRY(subtrees_and_finish(dVpos, dTriShade, n, nTop, nSub, subs, order, lo, hi, skey, top.leafMax, top.depthCap, top.pad, top.boxScale, top.maxDepth,
                             maxNodes, nodesF, nodes16, tris, trisRef, &nTotal, &maxDepth, stream));
  keepOutputs = true;
  out->nodes16 = nodes16, out->nodesF = nodesF, out->tris = tris, out->trisRef = trisRef;
  out->n_nodes = nTotal, out->maxDepth = maxDepth;
  cleanup();
  return hipSuccess;
}

// The EXACT device build (rt_options.bvh_builder = RT_BVH_DEVICE, and what RT_BVH_AUTO takes for big scenes): the host
// builder's split rules as kernels for the ranges above kSubMax triangles (k_top_*), then the exact subtrees, the rotation
// passes and the numbering every device build ends with.  `hSizeKey`: the host's size keys (rtbvh::planSceneExact).
hipError_t gpu_bvh_build_exact(const float* dVpos, const uint4* dTriShade, const float* hSizeKey, uint32_t n, const rtbvh::ScenePlan& P,
                               GpuBvh* out, hipStream_t stream) {
  *out = GpuBvh{};
  if (n == 0 || !hSizeKey) return hipErrorInvalidValue;
  const uint32_t leafMax = P.leafMax;
  const uint32_t maxItems = n / kSubMax + 2u;              // ranges of one level: disjoint, more than kSubMax triangles each
  const uint32_t maxBlk = n / kTopChunk + maxItems + 1u;   // their workgroups
  const uint32_t maxBinned = n / kSweepMaxD + 2u;
  const uint32_t maxSub = n / (leafMax + 1u) + 2u;         // parts hold more than leafMax triangles
  const uint32_t maxNodes = n;                             // a binary tree over n triangles, at least one per leaf
  Arena A;
  const size_t oLo = A.reserve((size_t)n * sizeof(float4)), oHi = A.reserve((size_t)n * sizeof(float4)), oKey = A.reserve((size_t)n * 4),
               oOrd = A.reserve((size_t)n * 4), oTmp = A.reserve((size_t)n * 4), oItems = A.reserve((size_t)maxItems * sizeof(TopItem)),
               oBlkItem = A.reserve((size_t)maxBlk * 4), oSweep = A.reserve((size_t)maxItems * 4), oIb = A.reserve((size_t)maxItems * 32),
               oPrep = A.reserve((size_t)maxItems * sizeof(TopPrep)), oDec = A.reserve((size_t)maxItems * sizeof(TopDec)),
               oMid = A.reserve((size_t)maxItems * 4), oRefs = A.reserve((size_t)maxItems * 8), oCbx = A.reserve((size_t)maxItems * 48),
               oBlockLeft = A.reserve((size_t)maxBlk * 4), oBins = A.reserve((size_t)maxBinned * 4u * 64u * 7u * 4u),
               oFlag = A.reserve(8), oSubs = A.reserve((size_t)maxSub * sizeof(SubItem)), oItems2 = A.reserve(sizeof(TopItem)),
               oDec2 = A.reserve(sizeof(TopDec)), oMid2 = A.reserve(8), oBlkItem2 = A.reserve((size_t)maxBlk * 4),
               oBlockLeft2 = A.reserve((size_t)maxBlk * 4), oHist = A.reserve(256 * 4);
  float4 *nodesF = nullptr, *tris = nullptr, *trisRef = nullptr;
  uint4* nodes16 = nullptr;
  bool keepOutputs = false;
  auto cleanup = [&]() {
    A.release();
    if (!keepOutputs)
      for (void* p : {(void*)nodes16, (void*)nodesF, (void*)tris, (void*)trisRef})
        if (p) (void)hipFree(p);
  };
  GB_TRY(A.commit());
  GB_TRY(hipMalloc((void**)&nodes16, 2 * (size_t)maxNodes * sizeof(uint4)));
  GB_TRY(hipMalloc((void**)&nodesF, 4 * (size_t)maxNodes * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&tris, 3 * (size_t)n * sizeof(float4)));
  GB_TRY(hipMalloc((void**)&trisRef, 3 * (size_t)n * sizeof(float4)));
  float4 *lo = A.at<float4>(oLo), *hi = A.at<float4>(oHi);
  float* skey = A.at<float>(oKey);
  uint32_t *ord = A.at<uint32_t>(oOrd), *tmp = A.at<uint32_t>(oTmp), *blkItem = A.at<uint32_t>(oBlkItem), *sweepList = A.at<uint32_t>(oSweep),
           *mid = A.at<uint32_t>(oMid), *blockLeft = A.at<uint32_t>(oBlockLeft), *flag = A.at<uint32_t>(oFlag);
  TopItem *items = A.at<TopItem>(oItems), *items2 = A.at<TopItem>(oItems2);
  TopDec* dec2 = A.at<TopDec>(oDec2);
  uint32_t *mid2 = A.at<uint32_t>(oMid2), *blkItem2 = A.at<uint32_t>(oBlkItem2), *blockLeft2 = A.at<uint32_t>(oBlockLeft2), *hist = A.at<uint32_t>(oHist);
  int *ib = A.at<int>(oIb), *cbx = A.at<int>(oCbx), *bins = A.at<int>(oBins);
  TopPrep* prep = A.at<TopPrep>(oPrep);
  TopDec* dec = A.at<TopDec>(oDec);
  int32_t* refs = A.at<int32_t>(oRefs);
  SubItem* subs = A.at<SubItem>(oSubs);
  RNode* nodes = reinterpret_cast<RNode*>(nodesF);

  const dim3 blk(256), grdN((n + 255) / 256);
  const bool verbose = getenv("RT_BVH_VERBOSE") != nullptr;
  const auto tB0 = std::chrono::steady_clock::now();
  auto msSince = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tB0).count(); };
  GB_TRY(hipMemcpyAsync(skey, hSizeKey, (size_t)n * 4, hipMemcpyHostToDevice, stream));
  GB_TRY(hipMemsetAsync(flag, 0, 8, stream));
  hipLaunchKernelGGL(k_tri_boxes, grdN, blk, 0, stream, dVpos, dTriShade, n, lo, hi);
  hipLaunchKernelGGL(k_iota, grdN, blk, 0, stream, ord, n);
  static std::atomic<unsigned long long> sweepLds{0};
  GB_TRY(grant_lds(reinterpret_cast<const void*>(&k_top_sweep), kSweepLds, sweepLds));

  struct HItem {
    uint32_t b, e, node;
  };
  std::vector<HItem> cur{{0u, n, 0u}}, next;
  std::vector<TopItem> hItems;
  std::vector<uint32_t> hBlk, hSweep, hMid;
  std::vector<int32_t> hRefs;
  uint32_t nTop = 1, nSub = 0, depth = 0, topMaxDepth = 0;
  const SubItem whole{0u, n, 0u, ~0u, 0u};
  if (n <= kSubMax) {  // the whole scene is one exact subtree: no top
    GB_TRY(hipMemcpyAsync(subs, &whole, sizeof whole, hipMemcpyHostToDevice, stream));
    GB_TRY(hipStreamSynchronize(stream));
    cur.clear(), nTop = 0, nSub = 1;
  }
  while (!cur.empty()) {
    const uint32_t count = (uint32_t)cur.size();
    if ((int)depth >= rtbvh::kMaxDepth - 1 || count > maxItems) {
      cleanup();
      return hipErrorInvalidValue;  // (cannot happen: the split choice enforces the depth budget)
    }
    hItems.resize(count), hBlk.clear(), hSweep.clear();
    uint32_t nBinned = 0;
    for (uint32_t r = 0; r < count; ++r) {
      const uint32_t sz = cur[r].e - cur[r].b;
      TopItem& it = hItems[r];
      it.b = cur[r].b, it.e = cur[r].e, it.blk0 = (uint32_t)hBlk.size(), it.node = cur[r].node, it.pad0 = it.pad1 = 0;
      it.nb = sz > kSweepMaxD ? (sz >= 65536u ? 64u : 16u) : 0u;
      it.binIdx = it.nb ? nBinned++ : (uint32_t)hSweep.size();
      if (!it.nb) hSweep.push_back(r);
      hBlk.insert(hBlk.end(), (sz + kTopChunk - 1u) / kTopChunk, r);
    }
    const uint32_t nBlk = (uint32_t)hBlk.size(), nSweep = (uint32_t)hSweep.size();
    if (nBlk > maxBlk || nBinned > maxBinned) {
      cleanup();
      return hipErrorInvalidValue;
    }
    GB_TRY(hipMemcpyAsync(items, hItems.data(), (size_t)count * sizeof(TopItem), hipMemcpyHostToDevice, stream));
    GB_TRY(hipMemcpyAsync(blkItem, hBlk.data(), (size_t)nBlk * 4, hipMemcpyHostToDevice, stream));
    if (nSweep) GB_TRY(hipMemcpyAsync(sweepList, hSweep.data(), (size_t)nSweep * 4, hipMemcpyHostToDevice, stream));
    const dim3 grdI((count + 255) / 256);
    hipLaunchKernelGGL(k_top_init, dim3((count * 12u + 255u) / 256u), blk, 0, stream, count, ib, cbx);
    hipLaunchKernelGGL(k_top_bounds, dim3(nBlk), blk, 0, stream, items, blkItem, ord, lo, hi, skey, ib);
    hipLaunchKernelGGL(k_top_prep, grdI, blk, 0, stream, items, count, ib, prep, bins);
    if (nBinned) {
      hipLaunchKernelGGL(k_top_bins, dim3(nBlk), blk, 0, stream, items, blkItem, ord, lo, hi, skey, prep, bins);
      hipLaunchKernelGGL(k_top_choose, dim3(count), dim3(64), 0, stream, items, count, depth, P.depthCap, leafMax, prep, bins, dec, mid, flag);
      hipLaunchKernelGGL(k_top_count, dim3(nBlk), blk, 0, stream, items, blkItem, ord, lo, hi, skey, dec, blockLeft);
      hipLaunchKernelGGL(k_top_offsets, dim3(count), dim3(64), 0, stream, items, blockLeft);
      hipLaunchKernelGGL(k_top_scatter, dim3(nBlk), blk, 0, stream, items, blkItem, ord, lo, hi, skey, dec, mid, blockLeft, tmp);
      hipLaunchKernelGGL(k_top_copyback, dim3(nBlk), blk, 0, stream, items, blkItem, tmp, ord);
    }
    if (nSweep)
      hipLaunchKernelGGL(k_top_sweep, dim3(nSweep), dim3(kSweepT), kSweepLds, stream, items, sweepList, depth, P.depthCap, leafMax, ib, lo, hi,
                         skey, ord, tmp, mid);
    hMid.resize(count);
    uint32_t hostFlag = 0;
    GB_TRY(hipMemcpyAsync(hMid.data(), mid, (size_t)count * 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipMemcpyAsync(&hostFlag, flag, 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    GB_TRY(hipGetLastError());
    if (hostFlag) {
      // Binned ranges the host would cut at its MEDIAN (no admissible SAH cut, or the lopsided-split guard): medianSplit is the
      // lower half in (centroid along the axis of the largest extent, id) order.  The pivot — the n/2-th smallest 64-bit key — by
      // radix selection, a byte per pass; then the same stable partition as the binned cuts, with "key < pivot" as its test.
      std::vector<TopDec> hDec(count);
      std::vector<int> hIb(8 * (size_t)count);
      GB_TRY(hipMemcpyAsync(hDec.data(), dec, (size_t)count * sizeof(TopDec), hipMemcpyDeviceToHost, stream));
      GB_TRY(hipMemcpyAsync(hIb.data(), ib, (size_t)count * 32, hipMemcpyDeviceToHost, stream));
      GB_TRY(hipMemsetAsync(flag, 0, 8, stream));
      GB_TRY(hipStreamSynchronize(stream));
      auto unkey = [](int i) {
        const int b = i >= 0 ? i : i ^ 0x7fffffff;
        float f;
        memcpy(&f, &b, 4);
        return f;
      };
      auto bitsf = [](uint32_t u) {
        float f;
        memcpy(&f, &u, 4);
        return f;
      };
      for (uint32_t r = 0; r < count; ++r) {
        if (!hItems[r].nb || hDec[r].axis != -1) continue;
        const uint32_t rb = hItems[r].b, re = hItems[r].e, rn = re - rb, rBlk = (rn + kTopChunk - 1u) / kTopChunk;
        float ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = unkey(hIb[8 * (size_t)r + 4 + a]) - unkey(hIb[8 * (size_t)r + a]);
        const int f0 = ext[1] > ext[0] ? 1 : 0, ax = ext[2] > ext[f0] ? 2 : f0;  // (std::sort's insertion sort of three: the first of equal extents)
        unsigned long long prefix = 0;
        uint32_t k = rn / 2u;  // (0-based rank of the pivot: exactly n / 2 keys lie below it — the keys are distinct)
        for (int shift = 56; shift >= 0; shift -= 8) {
          uint32_t hh[256];
          GB_TRY(hipMemsetAsync(hist, 0, 256 * 4, stream));
          const uint32_t g = (rn + 255u) / 256u < 1024u ? (rn + 255u) / 256u : 1024u;
          hipLaunchKernelGGL(k_top_select_hist, dim3(g), blk, 0, stream, ord, rb, re, ax, lo, hi, prefix, shift, hist);
          GB_TRY(hipMemcpyAsync(hh, hist, sizeof hh, hipMemcpyDeviceToHost, stream));
          GB_TRY(hipStreamSynchronize(stream));
          uint32_t j = 0;
          while (j < 255u && k >= hh[j]) k -= hh[j], ++j;
          prefix |= (unsigned long long)j << shift;
        }
        hDec[r] = TopDec{4 + ax, 0, bitsf((uint32_t)(prefix >> 32)), bitsf((uint32_t)prefix)};
        TopItem mi = hItems[r];
        mi.blk0 = 0;
        GB_TRY(hipMemcpyAsync(items2, &mi, sizeof mi, hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemcpyAsync(dec2, &hDec[r], sizeof(TopDec), hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemcpyAsync(mid2, &hMid[r], 4, hipMemcpyHostToDevice, stream));
        GB_TRY(hipMemsetAsync(blkItem2, 0, (size_t)rBlk * 4, stream));
        hipLaunchKernelGGL(k_top_count, dim3(rBlk), blk, 0, stream, items2, blkItem2, ord, lo, hi, skey, dec2, blockLeft2);
        hipLaunchKernelGGL(k_top_offsets, dim3(1), dim3(64), 0, stream, items2, blockLeft2);
        hipLaunchKernelGGL(k_top_scatter, dim3(rBlk), blk, 0, stream, items2, blkItem2, ord, lo, hi, skey, dec2, mid2, blockLeft2, tmp);
        hipLaunchKernelGGL(k_top_copyback, dim3(rBlk), blk, 0, stream, items2, blkItem2, tmp, ord);
        GB_TRY(hipStreamSynchronize(stream));  // (mi, hDec[r] and hMid[r] are read by the copies above)
      }
      GB_TRY(hipGetLastError());
    }
    // the children: leaves (<= leafMax), parts (<= kSubMax: one exact subtree each), or ranges of the next level
    next.clear(), hRefs.resize(2 * (size_t)count);
    for (uint32_t r = 0; r < count; ++r) {
      const uint32_t m = hMid[r];
      if (m <= cur[r].b || m >= cur[r].e) {
        cleanup();
        return hipErrorInvalidValue;
      }
      for (int c = 0; c < 2; ++c) {
        const uint32_t cb0 = c ? m : cur[r].b, ce0 = c ? cur[r].e : m, sz = ce0 - cb0;
        int32_t ref;
        if (sz <= leafMax) ref = rtbvh::encodeLeaf(cb0, sz), topMaxDepth = depth + 1u;
        else if (sz <= kSubMax) ref = (int32_t)~(rtbvh::kPartFlag | nSub++), topMaxDepth = depth + 1u;
        else ref = (int32_t)nTop, next.push_back(HItem{cb0, ce0, nTop++});
        hRefs[2 * (size_t)r + c] = ref;
      }
    }
    if (nSub > maxSub || nTop > maxNodes) {
      cleanup();
      return hipErrorInvalidValue;
    }
    GB_TRY(hipMemcpyAsync(refs, hRefs.data(), 2 * (size_t)count * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_top_childbox, dim3(nBlk), blk, 0, stream, items, blkItem, ord, lo, hi, mid, cbx);
    hipLaunchKernelGGL(k_top_emit, grdI, blk, 0, stream, items, count, depth, mid, refs, cbx, P.pad, ord, nodes, subs);
    // (the next level's uploads overwrite `items` and `refs` — and the host vectors they come from —: the kernels above must
    // have read them)
    GB_TRY(hipStreamSynchronize(stream));
    cur.swap(next);
    ++depth;
  }
  // (the top's packed records are written with everybody else's by rotate_and_pack: k_sub_relocate patches the float records)
  const double tTop = msSince();
  uint32_t nTotal = 0, maxDepth = 0;
  GB_TRY(subtrees_and_finish(dVpos, dTriShade, n, nTop, nSub, subs, ord, lo, hi, skey, leafMax, P.depthCap, P.pad, P.boxScale, topMaxDepth, maxNodes,
                             nodesF, nodes16, tris, trisRef, &nTotal, &maxDepth, stream));
  keepOutputs = true;
  out->nodes16 = nodes16, out->nodesF = nodesF, out->tris = tris, out->trisRef = trisRef;
  out->n_nodes = nTotal, out->maxDepth = maxDepth;
  const double tDone = msSince();
  cleanup();
  if (verbose)
    fprintf(stderr, "device build: top of %u nodes over %u parts in %u levels %.2f ms (allocations included), subtrees + rotations + records %.2f ms, release %.2f ms\n",
            nTop, nSub, depth, tTop, tDone - tTop, msSince() - tDone);
  return hipSuccess;
}

}  // namespace rtk
