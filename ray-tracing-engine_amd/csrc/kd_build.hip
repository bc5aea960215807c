// kd_build.hip — the photon kd-tree order on the device.
//
// The reference builds its photon tree by recursive std::nth_element
// (source/kdtree.h:60-69): the node of range [b,e) is the element the selection leaves
// at b + (e-b)/2, the axis cycles x,y,z.  The tree IS the resulting array order, and the
// reference's approximate k-NN (kdtree.h:87-107: heap seeded with the first k array
// elements, lagging m_bestdist) depends on that order down to how equal keys are arranged
// — photons on an axis-aligned wall tie exactly.  So this file restates libstdc++'s
// nth_element (bits/stl_algo.h: __introselect, __unguarded_partition_pivot,
// __move_median_to_first, __unguarded_partition, __insertion_sort, __heap_select) such
// that the ARRANGEMENT it leaves is the library's, element for element:
//   * one wave per range; all ranges of a tree level are independent (one launch per level);
//   * Hoare's partition pairs the i-th element from the left that is not < pivot with the
//     i-th from the right that is not > pivot while they have not crossed — an order-only
//     rule, so the wave applies it 64 + 64 elements at a time (ballot + rank, no atomics)
//     and hands the last < 192 elements to the sequential loop, run by one lane in LDS
//     from exactly the state (first, last) the sequential algorithm would be in;
//   * ranges of <= 192 elements run the whole selection sequentially in LDS.
// Checked against std::nth_element itself on the reference's golden photon list and on
// random inputs full of ties (tests/test_gpu_kdbuild.py).
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_kernels.h"

namespace rtk {

namespace {

constexpr int KD_WIN = 192;  // elements of the sequential window (LDS, per wave)
constexpr int KD_WAVES = 4;  // waves per workgroup

RT_DEV float kkey(const float4& v, int ax) { return ax == 0 ? v.x : ax == 1 ? v.y : v.z; }

RT_DEV void kd_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
RT_DEV void kd_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The sequential pieces of bits/stl_algo.h and bits/stl_heap.h on an array of items,
// compared by one coordinate (kdtree.h:33-39 node_cmp).  One lane runs these.
struct KdSeq {
  float4* a;
  int ax;
  RT_DEV bool lt(uint32_t i, uint32_t j) const { return kkey(a[i], ax) < kkey(a[j], ax); }
  RT_DEV void sw(uint32_t i, uint32_t j) const {
    const float4 t = a[i];
    a[i] = a[j], a[j] = t;
  }
  // stl_algo.h:79-98
  RT_DEV void move_median_to_first(uint32_t result, uint32_t ia, uint32_t ib, uint32_t ic) const {
    if (lt(ia, ib)) {
      if (lt(ib, ic)) sw(result, ib);
      else if (lt(ia, ic)) sw(result, ic);
      else sw(result, ia);
    } else if (lt(ia, ic)) sw(result, ia);
    else if (lt(ib, ic)) sw(result, ic);
    else sw(result, ib);
  }
  // stl_algo.h:1878-1895 (pivot is an element outside [first, last))
  RT_DEV uint32_t unguarded_partition(uint32_t first, uint32_t last, uint32_t pivot) const {
    const float p = kkey(a[pivot], ax);
    for (;;) {
      while (kkey(a[first], ax) < p) ++first;
      --last;
      while (p < kkey(a[last], ax)) --last;
      if (!(first < last)) return first;
      sw(first, last);
      ++first;
    }
  }
  // stl_algo.h:1799-1849
  RT_DEV void insertion_sort(uint32_t first, uint32_t last) const {
    if (first == last) return;
    for (uint32_t i = first + 1; i != last; ++i) {
      const float4 val = a[i];
      if (kkey(val, ax) < kkey(a[first], ax)) {
        for (uint32_t j = i; j > first; --j) a[j] = a[j - 1];  // move_backward(first, i, i + 1)
        a[first] = val;
      } else {
        uint32_t j = i;
        while (kkey(val, ax) < kkey(a[j - 1], ax)) a[j] = a[j - 1], --j;
        a[j] = val;
      }
    }
  }
  // stl_heap.h: __push_heap / __adjust_heap / __make_heap, stl_algo.h:1642-1650 __heap_select
  RT_DEV void push_heap(uint32_t first, int hole, int top, float4 val) const {
    int parent = (hole - 1) / 2;
    while (hole > top && kkey(a[first + parent], ax) < kkey(val, ax)) {
      a[first + hole] = a[first + parent];
      hole = parent;
      parent = (hole - 1) / 2;
    }
    a[first + hole] = val;
  }
  RT_DEV void adjust_heap(uint32_t first, int hole, int len, float4 val) const {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (kkey(a[first + child], ax) < kkey(a[first + child - 1], ax)) child--;
      a[first + hole] = a[first + child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      a[first + hole] = a[first + child - 1];
      hole = child - 1;
    }
    push_heap(first, hole, top, val);
  }
  RT_DEV void heap_select(uint32_t first, uint32_t middle, uint32_t last) const {
    const int len = (int)(middle - first);
    if (len >= 2)
      for (int parent = (len - 2) / 2;; parent--) {
        adjust_heap(first, parent, len, a[first + parent]);
        if (parent == 0) break;
      }
    for (uint32_t i = middle; i < last; ++i)
      if (kkey(a[i], ax) < kkey(a[first], ax)) {
        const float4 val = a[i];  // __pop_heap(first, middle, i)
        a[i] = a[first];
        adjust_heap(first, 0, len, val);
      }
  }
  // stl_algo.h:1964-1986
  RT_DEV void introselect(uint32_t first, uint32_t nth, uint32_t last, int depth) const {
    while (last - first > 3) {
      if (depth == 0) {
        heap_select(first, nth + 1, last);
        sw(first, nth);
        return;
      }
      --depth;
      const uint32_t mid = first + (last - first) / 2;
      move_median_to_first(first, first + 1, mid, last - 1);
      const uint32_t cut = unguarded_partition(first + 1, last, first);
      if (cut <= nth) first = cut;
      else last = cut;
    }
    insertion_sort(first, last);
  }
};

RT_DEV float4 shfl4(const float4& v, uint32_t src) {
  return make_float4(__shfl(v.x, (int)src, 64), __shfl(v.y, (int)src, 64), __shfl(v.z, (int)src, 64), __shfl(v.w, (int)src, 64));
}

// __unguarded_partition(lo, hi, pivot value p) by one wave.  Returns the cut (uniform).
RT_DEV uint32_t partition_wave(float4* a, uint32_t lo, uint32_t hi, float p, int ax, float4* win, uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t ua = lo, ub = hi, wL = 0, wR = 0;
  uint32_t lastL = lo - 1u, lastR = hi;  // positions of the last swap: the sequential state is (lastL + 1, lastR)
  uint64_t ML = 0, MR = 0;               // unconsumed stop positions of the live windows
  float4 vL = make_float4(0.f, 0.f, 0.f, 0.f), vR = vL;
  for (;;) {
    if (ML == 0) {
      if (ub - ua < 64u) break;
      wL = ua, ua += 64u;
      vL = a[wL + lane];
      ML = __ballot(!(kkey(vL, ax) < p));  // where `while (comp(first, pivot)) ++first` stops
    }
    if (MR == 0) {
      if (ub - ua < 64u) break;
      ub -= 64u, wR = ub;
      vR = a[wR + lane];
      MR = __ballot(!(p < kkey(vR, ax)));  // where `while (comp(pivot, last)) --last` stops
    }
    if (ML == 0 || MR == 0) continue;
    const uint32_t nL = (uint32_t)__popcll(ML), nR = (uint32_t)__popcll(MR), m = nL < nR ? nL : nR;
    const bool inL = (ML >> lane) & 1ull, inR = (MR >> lane) & 1ull;
    const uint32_t rL = (uint32_t)__popcll(ML & ((1ull << lane) - 1ull));  // i-th from the left
    const uint32_t rR = (uint32_t)__popcll((MR >> lane) >> 1);             // i-th from the right
    const bool giveL = inL && rL < m, giveR = inR && rR < m;
    if (giveR) part[rR] = lane;
    kd_wave_sync();
    const uint32_t pr = giveL ? part[rL] : lane;
    const float4 got = shfl4(vR, pr);  // (all lanes take part in the shuffle)
    if (giveL) {
      a[wL + lane] = got;   // iter_swap(first, last)
      a[wR + pr] = vL;
    }
    const uint64_t cL = __ballot(giveL), cR = __ballot(giveR);
    lastL = wL + (63u - (uint32_t)__clzll((long long)cL));
    lastR = wR + (uint32_t)(__ffsll((long long)cR) - 1);
    ML &= ~cL, MR &= ~cR;
    kd_wave_sync();
  }
  kd_fence();
  // the rest sequentially, from the state the sequential algorithm is in after these swaps:
  // everything left of the live left window is done, everything right of the live right one too
  const uint32_t liveL = ML ? wL : ua, liveR = MR ? wR + 64u : ub;
  const uint32_t f0 = lastL + 1u > liveL ? lastL + 1u : liveL;
  const uint32_t l0 = lastR < liveR ? lastR : liveR;
  const int T = (int)(l0 - f0);  // <= 191
  for (int i = (int)lane; i < T; i += 64) win[i] = a[f0 + (uint32_t)i];
  kd_wave_sync();
  int cutRel = 0;
  if (lane == 0) {
    int ff = 0, ll = T;
    for (;;) {
      while (ff < T && kkey(win[ff], ax) < p) ++ff;  // (position l0 holds an element that is not < p)
      --ll;
      while (ll >= 0 && p < kkey(win[ll], ax)) --ll;  // (position f0 - 1 holds one that is not > p)
      if (!(ff < ll)) break;
      const float4 t = win[ff];
      win[ff] = win[ll], win[ll] = t;
      ++ff;
    }
    cutRel = ff;
  }
  cutRel = __builtin_amdgcn_readfirstlane(cutRel);
  kd_wave_sync();
  for (int i = (int)lane; i < T; i += 64) a[f0 + (uint32_t)i] = win[i];
  kd_fence();
  return f0 + (uint32_t)cutRel;
}

// std::nth_element(a + first, a + nth, a + last, by coordinate ax) by one wave.
// depthOverride < 0: libstdc++'s 2 * lg(n).
RT_DEV void nth_element_wave(float4* a, uint32_t first, uint32_t nth, uint32_t last, int ax, int depthOverride, float4* win,
                             uint32_t* part) {
  const uint32_t lane = threadIdx.x & 63u;
  if (first == last || nth == last) return;
  int depth = depthOverride >= 0 ? depthOverride : 2 * (31 - __clz((int)(last - first)));
  for (;;) {
    const uint32_t size = last - first;
    if (size <= (uint32_t)KD_WIN) {
      // the remainder of the selection sequentially, in LDS
      for (uint32_t i = lane; i < size; i += 64u) win[i] = a[first + i];
      kd_wave_sync();
      if (lane == 0) KdSeq{win, ax}.introselect(0u, nth - first, size, depth);
      kd_wave_sync();
      for (uint32_t i = lane; i < size; i += 64u) a[first + i] = win[i];
      kd_fence();
      return;
    }
    if (depth == 0) {  // (rare on ranges this large) stl_algo.h:1970-1976
      if (lane == 0) {
        const KdSeq S{a, ax};
        S.heap_select(first, nth + 1u, last);
        S.sw(first, nth);
      }
      kd_fence();
      return;
    }
    --depth;
    const uint32_t mid = first + size / 2u;
    if (lane == 0) KdSeq{a, ax}.move_median_to_first(first, first + 1u, mid, last - 1u);
    kd_fence();
    const float p = kkey(a[first], ax);
    const uint32_t cut = partition_wave(a, first + 1u, last, p, ax, win, part);
    if (cut <= nth) first = cut;
    else last = cut;
  }
}

}  // namespace

// kdtree::make_tree (kdtree.h:60-69), all ranges of tree level `level`.
__global__ __launch_bounds__(64 * KD_WAVES) void k_kd_level(float4* __restrict__ items, uint32_t n, uint32_t level,
                                                            int depthOverride) {
  __shared__ float4 win[KD_WAVES][KD_WIN];
  __shared__ uint32_t part[KD_WAVES][64];
  const uint32_t wv = threadIdx.x >> 6;
  const uint64_t r = (uint64_t)blockIdx.x * KD_WAVES + wv;
  if (r >= (1ull << level)) return;
  uint32_t b = 0, e = n;
  for (int bit = (int)level - 1; bit >= 0; --bit) {
    if (e <= b) return;
    const uint32_t mid = b + (e - b) / 2u;
    if ((r >> bit) & 1ull) b = mid + 1u;
    else e = mid;
  }
  if (e <= b) return;
  nth_element_wave(items, b, b + (e - b) / 2u, e, (int)(level % 3u), depthOverride, win[wv], part[wv]);
}

// Stored photons of the emission kernel's slot arrays (k_emit: slot j holds at most one
// particle, flagged in pos.w), compacted in emission order: item = {position, slot}.
// One workgroup; *count receives the number of stored photons.
__global__ __launch_bounds__(1024) void k_photon_compact(const float4* __restrict__ slots, uint32_t n, float4* __restrict__ items,
                                                         uint32_t* __restrict__ count) {
  __shared__ uint32_t sums[1024];
  const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
  const uint32_t b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
  uint32_t c = 0;
  for (uint32_t j = b; j < e; ++j) c += slots[j].w != 0.f;
  sums[t] = c;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {  // inclusive scan
    const uint32_t v = t >= off ? sums[t - off] : 0u;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  uint32_t o = sums[t] - c;
  for (uint32_t j = b; j < e; ++j) {
    const float4 s = slots[j];
    if (s.w != 0.f) items[o++] = make_float4(s.x, s.y, s.z, __uint_as_float(j));
  }
  if (t == 1023u) *count = sums[1023];
}

// tree order -> the context's photon arrays: position, and income direction + weight of the
// emission slot the item came from
__global__ void k_photon_gather(const float4* __restrict__ items, const float4* __restrict__ slotDir, uint32_t n,
                                float4* __restrict__ phPos, float4* __restrict__ phDir, uint32_t* __restrict__ perm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 it = items[i];
  const uint32_t src = __float_as_uint(it.w);
  phPos[i] = make_float4(it.x, it.y, it.z, 0.f);
  if (phDir) phDir[i] = slotDir[src];
  if (perm) perm[i] = src;
}

// The tree's EXPLICIT topology, for the k-NN walk (rt_kernels.hip knn_query).  The reference's tree is implicit in the
// array order (kdtree.h:60-69: the node of a range [b, e) is its median element b + (e - b) / 2, the halves are its
// children, the axis cycles with the depth), and a walk that carries (b, e, depth) spends a third of its instructions on
// range and level arithmetic.  One 32-byte record per photon instead (one cache line, two requests): {position x, y, z,
// left child | axis << 30}{right child, the PARENT's split coordinate (float bits), the parent's axis, 0}; no child =
// KD_NONE.  The parent's split lets a far child that waited on the walk's stack be re-tested against kdtree.h:105
// (dx * dx >= m_bestdist) from its own record when it is popped.
__global__ void k_kd_topology(const float4* __restrict__ phPos, uint32_t n, uint4* __restrict__ topo) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t b = 0, e = n, level = 0, parent = 0x3fffffffu;
  for (;;) {
    const uint32_t m = b + (e - b) / 2;
    if (m == i) break;
    parent = m;
    if (i < m) e = m;
    else b = m + 1;
    ++level;
  }
  const uint32_t left = i > b ? b + (i - b) / 2 : 0x3fffffffu, right = e > i + 1 ? (i + 1) + (e - i - 1) / 2 : 0x3fffffffu;
  uint32_t psplit = 0, paxis = 0;
  if (parent != 0x3fffffffu) {
    paxis = (level - 1) % 3;
    const float4 pp = phPos[parent];
    psplit = __float_as_uint(paxis == 0 ? pp.x : paxis == 1 ? pp.y : pp.z);
  }
  const float4 me = phPos[i];
  topo[2 * (size_t)i] = make_uint4(__float_as_uint(me.x), __float_as_uint(me.y), __float_as_uint(me.z), left | ((level % 3) << 30));
  topo[2 * (size_t)i + 1] = make_uint4(right, psplit, paxis, 0u);
}

hipError_t launch_kd_topology(const float4* phPos, uint32_t n, uint4* topo, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_kd_topology, dim3((n + 255) / 256), dim3(256), 0, stream, phPos, n, topo);
  return hipGetLastError();
}

hipError_t launch_kd_build(float4* items, uint32_t n, int depthOverride, hipStream_t stream) {
  if (n < 2) return hipSuccess;
  // levels: ranges halve until they are empty; level L has at most 2^L ranges
  for (uint32_t level = 0; level < 32u; ++level) {
    if ((n >> level) == 0u) break;  // every range of this level is empty
    const uint64_t ranges = 1ull << level;
    const uint32_t blocks = (uint32_t)((ranges + KD_WAVES - 1) / KD_WAVES);
    hipLaunchKernelGGL(k_kd_level, dim3(blocks), dim3(64 * KD_WAVES), 0, stream, items, n, level, depthOverride);
  }
  return hipGetLastError();
}

hipError_t launch_photon_compact(const float4* slots, uint32_t n, float4* items, uint32_t* count, hipStream_t stream) {
  hipLaunchKernelGGL(k_photon_compact, dim3(1), dim3(1024), 0, stream, slots, n, items, count);
  return hipGetLastError();
}

hipError_t launch_photon_gather(const float4* items, const float4* slotDir, uint32_t n, float4* phPos, float4* phDir,
                                uint32_t* perm, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_photon_gather, dim3((n + 255) / 256), dim3(256), 0, stream, items, slotDir, n, phPos, phDir, perm);
  return hipGetLastError();
}

}  // namespace rtk
