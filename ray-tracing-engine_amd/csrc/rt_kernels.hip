// rt_kernels.hip — gfx950 kernels of the path-tracing hot path.
//
// Execution model (wave64, CDNA4):
//   * one lane = one (pixel, sample) pair in every render kernel: a wave integrates
//     S = 1 << sshift consecutive samples of P = 64 >> sshift pixels side by side and the
//     pixel's owner lane adds them in sample order, so every pixel's float sum is formed
//     in exactly the order the reference forms it (Renderer.cpp:219-258) whatever S, the
//     number of launches or the number of GPUs the frame is split over;
//   * BVH traversal keeps a per-lane stack in LDS laid out [level][thread]
//     (ds_read/write_b32, lane-contiguous => bank-conflict free), node records are
//     32 B (12 binary16 box planes + 2 child refs: 2 x dwordx4 per lane per step),
//     triangle records 48 B (3 x dwordx4);
//   * the photon k-NN keeps its k-heap in LDS ([slot][thread]) and re-uses the
//     traversal stack region for the kd-tree's per-level split distances;
//   * the pooled integrator (BVH direct lighting) runs as ONE persistent workgroup of 16
//     waves per CU (fewer when deep stacks fill the LDS): its LDS holds every wave's private
//     stack and ray pool plus the whole BVH when it fits, else its most-visited prefix;
//     waves draw wave tiles from a global counter; after the tree copy no wave ever
//     synchronises with another (wave-scope fences only), counters leave through one atomic
//     per wave;
//   * a wave issues one instruction of ANY kind per 4 cycles, so the hot loops are written for
//     instruction COUNT: wave-uniform loops around predicated regions, lane masks made by hand
//     (ballot / inverse ballot + scalar bit operations) where the compiler would add compares,
//     and s_setprio by phase (the descent's dependent LDS round trips go first).
// Built with -ffp-contract=off (bit parity with the x86-64 oracle, no FMA).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "bvh_build.h"
#include "rt_device.h"
#include "rt_kernels.h"

using namespace rtd;

namespace rtk {

constexpr int BLOCK = 64;
constexpr int STACK = rtbvh::kMaxDepth;  // 32 words per lane
constexpr int KMAX = RTK_KMAX;           // photon heap capacity per lane

struct Lds {
  uint32_t* stack;  // [STACK][BLOCK] this thread's column: stack[level * BLOCK]
  uint2* heap;      // [KMAX][BLOCK] {distance bits, photon index}
};

// Diagnostic build only (-DRT_PHASE_TIMING, tools/phase_timing.sh): wave-level shader-clock
// time per section of the pooled k_render, accumulated in LDS by lane 0 at wave-uniform
// points and flushed to counters[16..].  Never compiled into the product library.
#ifdef RT_PHASE_TIMING
__shared__ unsigned long long g_phAcc[24];
__shared__ unsigned long long g_phT0;
#define PH(p)                                                  \
  do {                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (threadIdx.x == 0) g_phAcc[p] += t_ - g_phT0, g_phT0 = t_; \
  } while (0)
#define PHC(p) do { if (threadIdx.x == 0) g_phAcc[p] += 1; } while (0)
#else
#define PH(p) do { } while (0)
#define PHC(p) do { } while (0)
#endif
enum { PH_SETUP = 0, PH_PRIMARY, PH_VSETUP, PH_FILL, PH_HANDOUT, PH_STEAL, PH_DESCENT, PH_LEAF, PH_POOLMISC, PH_BSDF,
       PH_ACCUM, PH_N_ROUNDS, PH_N_STEPS, PH_N_POOLS, PH_TAIL, PH_FOREIGN, PH_COUNT };

struct LaneStats {
  uint32_t closest = 0, shadow = 0, knn = 0, nodes = 0, tris = 0, kd = 0;
  uint32_t wnode = 0, wleaf = 0;  // wave-level node steps / leaf phases (counted by the first active lane)
  uint32_t lwait = 0, lidle = 0;  // lanes holding a leaf / no ray during those node steps
  uint32_t frames = 0;            // Q8: block frames fetched
};

struct HitRec {
  float t, u, v;
  uint32_t id, mesh;
};

RT_DEV float safe_inv(float d) {
  // finite stand-in for 1/0 so that lo * inv - o * inv never forms 0 * inf.  The
  // reciprocal only feeds the (padded, conservative) slab test, never a reference
  // expression, so the 1-ulp hardware v_rcp_f32 is enough: its error is four orders
  // below the box padding in t units.
  // (clamped, not selected: rcp(+-0) and rcp(denormal) are +-inf, med3 brings them to +-1e20 —
  // two instructions instead of compare + copysign + select)
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d), -1e20f, 1e20f);
}

RT_DEV f3 f4xyz(const float4& a) { return mk(a.x, a.y, a.z); }

// Conservative slab test of one padded child box.  Returns entry distance in tn.
// t = lo*inv - o*inv as ONE fma per plane (this is our own box arithmetic, not a
// reference expression, so contraction is allowed here): its absolute error is
// ~ulp(o*inv) = 6e-8*|o|*|inv|, three orders below the box padding expressed in t
// units (pad*|inv|, pad >= 6e-5*|o|max), so no padded box is ever wrongly culled.
RT_DEV float h2f_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
RT_DEV float h2f_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
// byte k of a word as a float: the compiler selects v_cvt_f32_ubyteK (extract + convert, one instruction)
RT_DEV float ub0(uint32_t w) { return (float)(w & 0xffu); }
RT_DEV float ub1(uint32_t w) { return (float)((w >> 8) & 0xffu); }
RT_DEV float ub2(uint32_t w) { return (float)((w >> 16) & 0xffu); }
RT_DEV float ub3(uint32_t w) { return (float)(w >> 24); }

RT_DEV bool slab(float lx, float ly, float lz, float hx, float hy, float hz, f3 inv, f3 oi, float tmax,
                 float& tn) {
  float ax = __builtin_fmaf(lx, inv.x, -oi.x), bx = __builtin_fmaf(hx, inv.x, -oi.x);
  float ay = __builtin_fmaf(ly, inv.y, -oi.y), by = __builtin_fmaf(hy, inv.y, -oi.y);
  float az = __builtin_fmaf(lz, inv.z, -oi.z), bz = __builtin_fmaf(hz, inv.z, -oi.z);
  float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
  float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  tn = tnear;
  return tnear <= fminf(tfar, tmax);
}

// The same test with the planes already ordered along the ray: `n*` = the plane the ray
// crosses first on that axis (lo when the direction component is >= 0, else hi), `f*` the
// other one.  t is monotone in the plane coordinate (one correctly rounded fma, inv fixed and
// finite: safe_inv), so near/far selected by the sign of inv ARE min/max of the pair — same
// tnear, tfar, same decision, six instructions fewer per child pair.
RT_DEV bool slab_ordered(float nx, float fx, float ny, float fy, float nz, float fz, f3 inv, f3 oi, float tmax, float& tn) {
  const float ax = __builtin_fmaf(nx, inv.x, -oi.x), bx = __builtin_fmaf(fx, inv.x, -oi.x);
  const float ay = __builtin_fmaf(ny, inv.y, -oi.y), by = __builtin_fmaf(fy, inv.y, -oi.y);
  const float az = __builtin_fmaf(nz, inv.z, -oi.z), bz = __builtin_fmaf(fz, inv.z, -oi.z);
  const float tnear = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.f));
  const float tfar = fminf(fminf(bx, by), bz);
  tn = tnear;
  return tnear <= fminf(tfar, tmax);
}
// (lo, hi) f16 pair -> (near, far): rotate by 16 when the ray runs against the axis
RT_DEV uint32_t order_planes(uint32_t w, uint32_t rot) { return __builtin_amdgcn_alignbit(w, w, rot); }

// lane mask of a predicate (HIP's __ballot(int) goes through a 0/1 register and a compare where
// the compiler does not fold it: two vector instructions per use)
RT_DEV uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr int32_t TERM = (int32_t)0x80000000;  // "this lane holds no live ray"

// Every LDS exchange in this file is between lanes of ONE wave (a wave owns its stack,
// heap and pool regions), so no s_barrier is ever needed: a wave's DS operations are
// executed in issue order, and this keeps the compiler from moving them across the
// hand-over point.  (With one wave per workgroup __syncthreads() compiled to the same
// thing; the persistent kernel runs 16 waves per workgroup, where it would not.)
RT_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// byte offset of g_lds inside the workgroup's LDS segment (static __shared__ objects of
// the diagnostic build come first)
RT_DEV uint32_t ldsNodeBase();

// Dynamic LDS of the render kernels.  The persistent pooled kernel keeps the TOP of the
// BVH here — node records [0, topK), which bvh_build lays out most-visited first — in
// front of the per-wave regions: one copy per CU, shared by all of its waves.
extern __shared__ uint32_t g_lds[];
RT_DEV uint32_t ldsNodeBase() { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)g_lds; }

// One lane's ray in flight.  round() advances every lane of the wave that holds a
// live ray by one "while-while" round: first the wave descends inner nodes until only a
// few lanes are still descending (lanes that already hold a leaf sit out, masked; the
// stragglers sit out the leaf phase), then all held leaves are intersected together —
// each wave iteration runs one kind of work instead of the union of both (the
// single-loop form measured 34 % lane utilisation on incoherent rays).  Between rounds
// a caller may hand finished lanes new rays or parts of other lanes' rays (vertex pool
// below).
constexpr int TRAV_CLOSEST = 0, TRAV_ANY = 1, TRAV_MIXED = 2;  // MIXED: per-lane `anyHit` flag

// LT: where node records live.  LT_NONE: HBM/L2 through the vector L1; LT_TOP: records
// [0, S.topK) in LDS (g_lds), the rest in HBM/L2; LT_ALL: the whole tree in LDS.
constexpr int LT_NONE = 0, LT_TOP = 1, LT_ALL = 2;
// + LT_Q8: the ONE-REQUEST node (rtbvh::Slot16, bvh_build.h): a 16-byte record — 12 8-bit box planes in the frame of the
// record's 16-KiB block + one word that locates both children — in ONE array with the triangle records.  A visit is one
// dwordx4 instead of two; a lane that enters a record of another block fetches that block's frame {origin, step} with it
// (one more request, in flight together with the record) and rebuilds its two slab constants: t = q * (step / d) +
// (origin - o) / d, so the 8-bit plane goes into the same fma the binary16 plane went into (v_cvt_f32_ubyteN extracts
// and converts in one instruction, v_perm_b32 orders a word's two (lo, hi) pairs along the ray).
constexpr int LT_Q8 = 4;
// Wave priority by phase (s_setprio): a wave in the DESCENT (a chain of dependent LDS round trips
// and short instruction runs) or in the pool bookkeeping issues ahead of waves that stream through
// shading arithmetic or triangle tests, instead of queueing behind them by age.  Measured: C2
// +5.1 %, C4 +2.2 % (descent 2, pool 1, leaf and shading 0; 1..3 in the descent are alike, a raised
// LEAF phase costs 2 %); on the 1 M-triangle scene, where the waves wait on memory rather than on
// each other, it costs 1 % — LT_NOPRIO (trees beyond kPrioMaxNodes) leaves everything at 0.
constexpr int LT_NOPRIO = 8;
// Compact ray pool (scenes whose stacks fill the LDS): per light the pool keeps the two
// parameters of the light sample instead of the direction to it — 2 words instead of 3 per lane and
// light, 768 B per wave with three lights — and the direction is rebuilt where it is needed by the
// arithmetic light_sample() performs (LightSource.h:46-49, the same operations in the same order:
// the same bits).  On the 1 M-triangle scene that is the sixteenth wave.
constexpr int LT_COMPACT = 16;
// ... and without the bounce directions either (another 768 B per wave; implies LT_COMPACT): every
// bounce ray is handed to its OWN pixel lane in the first hand-out and starts from the register that
// holds the direction anyway; a lane that steals part of a bounce ray takes the direction from the
// victim's registers.  The eight-million-triangle scene's sixteenth wave.
constexpr int LT_COMPACT2 = 32;
// ... and without the light samples' parameters or the vertex positions either (implies LT_COMPACT2): the pixel lane keeps its
// vertex and its RNG state from before its light draws in registers, a worker lane fetches both by lane shuffle and re-draws
// the light sample from the state — the same engine calls, the same bits.  168 words instead of 744 (the bounce keys, the
// hand-out list, the result bits): with 28 stack rows that is 7,840 B per wave — five workgroups of four waves fit a CU's LDS,
// which is handed out in 1,280-byte units —, TWENTY waves per CU for trees of up to 27 levels, and the instance is compiled
// for five waves per SIMD (96 VGPRs).
constexpr int LT_COMPACT3 = 64;
// 1 / det of the triangle test by rtd::recip_fast (3 instructions, the division's bits for |det| < 2^100) instead of the
// division: only the instances the launcher picks when the host has bounded |det| for the launch's rays (DevScene::slowRecip == 0).
constexpr int LT_FASTDET = 256;
constexpr uint32_t kPrioMaxNodes = 65536;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_u4_ptr;

template <int MODE, int LTX = LT_NONE>
struct Trav {
  static constexpr int LT = LTX & 3;
  static constexpr bool PRIO = (LTX & LT_NOPRIO) == 0;
  static constexpr bool DIVIDE = (LTX & LT_FASTDET) == 0;
  static constexpr bool Q8 = (LTX & LT_Q8) != 0;
  static_assert(!Q8 || LT == LT_NONE, "Q8 records are fetched from HBM / L2");
  // any-hit rays enter child 0 (the smaller box) first: the big-scene instances and the whole-tree-in-LDS ones (measured:
  // C5 -2.5 %, C5x8 -1.9 %, C2 -0.7 %, C1 -3 %; the partial-top instance of C4 loses 0.7 % to the extra scalar op and keeps
  // the distance order)
  static constexpr bool ANYORD = (LTX & LT_NOPRIO) != 0 || LT == LT_ALL;
  f3 o, d, inv, oi;
  uint32_t rotX, rotY, rotZ;  // 16 where the direction component is negative (order_planes); Q8: the v_perm_b32 selectors of the three plane words
  // Q8: 1 / d (inv and oi then hold the slab constants IN THE HELD FRAME: step / d and (o - origin) / d) and the block
  // whose frame that is
  f3 inv0;
  uint32_t fblk;
  float best;
  uint32_t bestId;
  bool found, anyHit;
  // LDS stack of this lane: row 0 holds a TERM sentinel, entries live in rows 1..depth;
  // `top` points at the top entry (or at the sentinel: an empty stack pops TERM by itself,
  // so no step ever tests for emptiness or clamps an index)
  uint32_t* top;
  uint32_t* base;
  int32_t cur;
  HitRec hit;
  // pool mode (TRAV_MIXED): several lanes may walk disjoint subtrees of ONE ray
  // (vertex_pool: work stealing).  `stolen` = bottom stack entries handed to other
  // lanes; closest-hit candidates are published to shared[0..1] = 64-bit key
  // (t bits << 32 | triangle id: unsigned min == closest t, lowest id on ties —
  // RayTracer.h:40's rule); the pixel lane recomputes u, v of the winner.
  int stolen;
  bool shared;  // this ray is (or was) walked by more than one lane
  unsigned long long* sharedKey;
  uint32_t pj;

  // entries on the stack (live lanes only: top >= base there, so the byte distance is unsigned —
  // a signed 64-bit pointer difference and division cost seven instructions)
  RT_DEV int depth() const {
    return (int)(((uint32_t)(uintptr_t)top - (uint32_t)(uintptr_t)base) / (4u * BLOCK));
  }
  RT_DEV void idle(uint32_t* stack) { cur = TERM, found = false, base = top = stack, stolen = 0, shared = false; }
  RT_DEV uint32_t peek() const { return *top; }
  RT_DEV void pop() { top -= BLOCK; }
  RT_DEV void start(f3 o_, f3 d_, float invScale) {
    o = o_, d = d_;
    const f3 i1 = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    if (Q8) {
      inv0 = i1, fblk = ~0u;  // (no frame held: the first step fetches the root block's)
      inv = oi = mk(0.f, 0.f, 0.f);
      // a plane word holds two (lo, hi) byte pairs — (x, y) of child 0, (z of child 0, x of child 1), (y, z) of child 1 —
      // and v_perm_b32 swaps the bytes of a pair whose axis the ray runs against
      const uint32_t nx = __float_as_uint(i1.x) >> 31, ny = __float_as_uint(i1.y) >> 31, nz = __float_as_uint(i1.z) >> 31;
      rotX = 0x03020100u ^ (nx * 0x00000101u) ^ (ny * 0x01010000u);
      rotY = 0x03020100u ^ (nz * 0x00000101u) ^ (nx * 0x01010000u);
      rotZ = 0x03020100u ^ (ny * 0x00000101u) ^ (nz * 0x01010000u);
    } else {
      oi = mk(o.x * i1.x, o.y * i1.y, o.z * i1.z);
      // boxes are stored as coordinate * boxScale (a power of two): fold 1/boxScale in
      inv = mk(i1.x * invScale, i1.y * invScale, i1.z * invScale);
      rotX = (__float_as_uint(i1.x) >> 31) << 4, rotY = (__float_as_uint(i1.y) >> 31) << 4, rotZ = (__float_as_uint(i1.z) >> 31) << 4;
    }
    best = 3.402823466e+38f;  // numeric_limits<float>::max(), RayTracer.h:30
    bestId = 0, found = false, top = base, cur = Q8 ? (int32_t)rtbvh::kQ8RootOffset : 0, stolen = 0, shared = false;
    *base = (uint32_t)TERM;
    // A ray with a NaN component cannot hit anything: Ray.cpp:9-24 then yields NaN u
    // or v for every triangle and every comparison fails (hemisphere samples are NaN
    // with p ~ 3e-8).  The slab test, built from min/max that drop
    // NaNs, would instead accept every box: ONE such ray walks all 500k nodes of the
    // 1M-triangle scene (measured: +0.6 s on a 50 ms frame).  Same result, no walk:
    // (three two-operand unordered tests)
    if (__builtin_isunordered(o.x, o.y) | __builtin_isunordered(o.z, d.x) | __builtin_isunordered(d.y, d.z)) cur = TERM;
  }
  RT_DEV bool live() const { return cur != TERM; }

  template <bool STATS>
  RT_DEV void round(const DevScene& S, LaneStats& st) {
    const int live0 = __popcll(wave_ballot(cur != TERM));
    // (at least 1: the loop below leaves when FEWER lanes than this descend, so 0 would never leave)
    const int exitBelow = max(1, min((int)S.leafT, (live0 * (int)S.leafMul) >> 6));
    uint32_t statWait = 0, statIdle = 0;
    if (STATS) statWait = (uint32_t)__popcll(wave_ballot(cur < 0 && cur != TERM)), statIdle = (uint32_t)__popcll(wave_ballot(cur == TERM));
    PHC(PH_N_ROUNDS);
    // The loop is WAVE-UNIFORM (a ballot decides, every lane leaves together) and the step a
    // plain predicated region inside it: with the lane condition as the loop condition the
    // compiler keeps per-lane exit masks (20 scalar instructions + 4 branches per step, and a
    // wave issues one instruction of ANY kind per 4 cycles).
    // (the lanes of a step are the ballot the loop control has just counted: the mask goes
    // straight back into exec — no second compare, no vector-to-scalar hand-over at the loop top)
    uint64_t inner = __builtin_amdgcn_ballot_w64(cur >= 0);
    // Any-hit rays (Renderer.cpp:54 only uses the bool) do not need the nearer child first: where ANYORD is set they
    // enter child 0 whenever it is hit — the builders put the SMALLER box there, the likelier place to find an occluder
    // close to the ray.  Same booleans, fewer visits: 1 M triangles 34.96 -> 33.54 nodes and 4.04 -> 3.50 triangle tests
    // per ray, 357 -> 348 ms; 8 M triangles 58.7 -> 57.6 ms; C2 3.17 -> 2.99 triangle tests per ray, 50.9 -> 50.6 ms.
    // (Denser-subtree-first, larger-box-first and the RTSAH rule lose: profiles/r03_anyhit_order_ab.txt.)
    const uint64_t anyFirst = !ANYORD ? 0ull : MODE == TRAV_ANY ? ~0ull : MODE == TRAV_MIXED ? __builtin_amdgcn_ballot_w64(anyHit) : 0ull;
    if (PRIO) __builtin_amdgcn_s_setprio(2);
    if (inner) for (;;) {
     if (__builtin_amdgcn_inverse_ballot_w64(inner)) {
      {
      float t0, t1;
      bool h0, h1;
      int2 ch;
      int32_t below;
      if (STATS) {
        st.nodes++;
        if (__ffsll((long long)wave_ballot(true)) - 1 == (int)(threadIdx.x & 63)) st.wnode++, st.lwait += statWait, st.lidle += statIdle;
      }
      if constexpr (Q8) {
        // ONE request: the 16-byte record; plus its block's frame when the lane last decoded a record of another block
        const char* const q8 = reinterpret_cast<const char*>(S.q8);
        const uint4 r = *reinterpret_cast<const uint4*>(q8 + (uint32_t)cur);
        const uint32_t blk = (uint32_t)cur >> S.q8ShiftBytes;
        if (blk != fblk) {
          const float4 fr = *reinterpret_cast<const float4*>(q8 + (blk << S.q8ShiftBytes));
          inv = mk(inv0.x * fr.w, inv0.y * fr.w, inv0.z * fr.w);
          oi = mk((o.x - fr.x) * inv0.x, (o.y - fr.y) * inv0.y, (o.z - fr.z) * inv0.z);
          fblk = blk;
          if (STATS) st.frames++;
        }
        PHC(PH_N_STEPS);
        below = (int32_t)peek();
        // both children's items are adjacent, child 0's first: one word locates them (bvh_build.h Slot16)
        const uint32_t ref = r.w, base0 = ref & ~15u;
        const uint32_t base1 = base0 + ((ref & 1u) ? ((ref & 4u) ? 96u : 48u) : 16u);
        ch.x = (ref & 1u) ? (int32_t)~(base0 | ((ref >> 2) & 1u)) : (int32_t)base0;
        ch.y = (ref & 2u) ? (int32_t)~(base1 | ((ref >> 3) & 1u)) : (int32_t)base1;
        const uint32_t w0 = __builtin_amdgcn_perm(r.x, r.x, rotX), w1 = __builtin_amdgcn_perm(r.y, r.y, rotY), w2 = __builtin_amdgcn_perm(r.z, r.z, rotZ);
        h0 = slab_ordered(ub0(w0), ub1(w0), ub2(w0),
                          ub3(w0), ub0(w1), ub1(w1), inv, oi, best, t0);
        h1 = slab_ordered(ub2(w1), ub3(w1), ub0(w2),
                          ub1(w2), ub2(w2), ub3(w2), inv, oi, best, t1);
      } else {
      // 32-B packed node: 12 x f16 planes + 2 refs (32-bit byte offset from a uniform base:
      // the load takes the base from SGPRs)
      uint4 a, b;
      // (`cur` of an inner node is the byte offset of its record)
      if (LT == LT_ALL || (LT == LT_TOP && (uint32_t)cur < S.topK * 32u)) {
        // from LDS (the tree copy starts at LDS byte offset 0 of the dynamic segment): a
        // divergent 32-B read costs the LDS a few cycles per wave where the vector L1 spends
        // one tag lookup per lane and dwordx4
        uint32_t off = (uint32_t)cur;
        if (LT == LT_TOP) asm volatile("" : "+v"(off));  // keep the two address spaces on separate paths (else: one flat load)
#ifdef RT_PHASE_TIMING
        lds_u4_ptr n = (lds_u4_ptr)(uintptr_t)(off + ldsNodeBase());
#else
        // (no static LDS in the product build: the dynamic segment, i.e. the tree copy, starts at LDS
        // address 0 — checked once on the host, launch_render2 — so a node's offset IS its address)
        lds_u4_ptr n = (lds_u4_ptr)(uintptr_t)off;
#endif
        const u32x4 va = n[0], vb = n[1];
        a = make_uint4(va.x, va.y, va.z, va.w), b = make_uint4(vb.x, vb.y, vb.z, vb.w);
      } else {
        const uint4* n = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(S.nodes) + (uint32_t)cur);
        a = n[0], b = n[1];
      }
      PHC(PH_N_STEPS);
      // the entry a miss would pop, fetched with the node (its latency hides behind the
      // node's; a read in the divergent pop branch made every step wait for LDS)
      below = (int32_t)peek();
      ch = make_int2((int)b.z, (int)b.w);
      const uint32_t x0 = order_planes(a.x, rotX), y0 = order_planes(a.y, rotY), z0 = order_planes(a.z, rotZ);
      const uint32_t x1 = order_planes(a.w, rotX), y1 = order_planes(b.x, rotY), z1 = order_planes(b.y, rotZ);
      h0 = slab_ordered(h2f_lo(x0), h2f_hi(x0), h2f_lo(y0), h2f_hi(y0), h2f_lo(z0), h2f_hi(z0), inv, oi, best, t0);
      h1 = slab_ordered(h2f_lo(x1), h2f_hi(x1), h2f_lo(y1), h2f_hi(y1), h2f_lo(z1), h2f_hi(z1), inv, oi, best, t1);
      }
      // select-based step: one divergent branch (the pop) instead of a four-way chain;
      // the far child is stored unconditionally (the slot is simply not claimed unless
      // both children were hit)
      // which child to enter: child 1 unless child 0 is hit and not farther.  Lane masks by hand
      // (ballots and scalar bit operations: three compares, and / and-not / or / and) — left to
      // the compiler the negation becomes a fourth compare.
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(h0), m1 = __builtin_amdgcn_ballot_w64(h1);
      const uint64_t mle = ANYORD ? (__builtin_amdgcn_ballot_w64(t0 <= t1) | anyFirst) : __builtin_amdgcn_ballot_w64(t0 <= t1);
      const bool takeY = __builtin_amdgcn_inverse_ballot_w64(m1 & ~(m0 & mle));
      const bool both = __builtin_amdgcn_inverse_ballot_w64(m0 & m1), any = __builtin_amdgcn_inverse_ballot_w64(m0 | m1);
      top[BLOCK] = (uint32_t)(takeY ? ch.x : ch.y);
      cur = any ? (takeY ? ch.y : ch.x) : below;
      top += both ? BLOCK : any ? 0 : -BLOCK;  // (a lane that pops the sentinel is dead until start())
      // leave the descent early once only a few lanes are still descending: they
      // sit out one leaf phase (masked) instead of making everyone else wait for them
      // (only when the lanes that would otherwise wait clearly outnumber them: in the
      // tail of a pool, with a handful of live rays, a round must not shrink to one step)
      // (desc < leafT and desc < live0 * leafMul / 64, folded into one threshold)
      }
     }
      inner = __builtin_amdgcn_ballot_w64(cur >= 0);
      if (__popcll(inner) < exitBelow) break;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    PH(PH_DESCENT);
    if (cur < 0 && cur != TERM) {
      const uint32_t code = ~(uint32_t)cur;
      // (byte offset of the leaf's first record | count - 1: the loads take the base from SGPRs)
      const uint32_t cnt = (code & 7u) + 1u;
      const char* const leaf = reinterpret_cast<const char*>(Q8 ? reinterpret_cast<const float4*>(S.q8) : S.tris) + (code & ~7u);
      if (STATS) st.wleaf += (uint32_t)(__ffsll((long long)wave_ballot(true)) - 1 == (int)(threadIdx.x & 63));
      // leaves hold 1..leaf_max (default 2) records: the first two are tested in
      // straight-line code with both records' loads in flight together
      const float4* r = reinterpret_cast<const float4*>(leaf);
      const uint32_t second = cnt > 1 ? 3u : 0u;  // a 1-triangle leaf re-reads its only record
      const float4 a0 = r[0], a1 = r[1], a2 = r[2];
      const float4 b0 = r[second + 0], b1 = r[second + 1], b2 = r[second + 2];
      bool stop = test_pair<STATS>(a0, a1, a2, b0, b1, b2, cnt > 1, st);
      for (uint32_t i = 2; i < cnt && !stop; i++) {
        const float4* q = reinterpret_cast<const float4*>(leaf + 48u * i);
        stop = test_record<STATS>(q[0], q[1], q[2], st);
      }
      cur = stop ? TERM : (int32_t)peek();
      pop();
    }
    if (PRIO && MODE == TRAV_MIXED) __builtin_amdgcn_s_setprio(1);  // back in the pool loop
    PH(PH_LEAF);
  }

  // The first two records of a leaf at once and without a branch: both tests run, every
  // acceptance is a select (RayTracer.h:40's rule applied to A, then to B), so the leaf
  // phase stays one basic block instead of a dozen exec-mask regions.  `two` = the leaf
  // holds a second record (else B is a re-read of A and is ignored).
  template <bool STATS>
  RT_DEV bool test_pair(const float4& p0, const float4& p1, const float4& p2, const float4& q0, const float4& q1,
                        const float4& q2, bool two, LaneStats& st) {
    if (STATS) st.tris += two ? 2u : 1u;
    float ua, va, ta, ub, vb, tb;
    const bool ha = tri_test(o, d, mk(p0.x, p0.y, p0.z), mk(p0.w, p1.x, p1.y), mk(p1.z, p1.w, p2.x), ua, va, ta, DIVIDE) && ta > 0.f;
    const bool hb = tri_test(o, d, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), ub, vb, tb, DIVIDE) && tb > 0.f && two;
    const bool isAny = MODE == TRAV_ANY || (MODE == TRAV_MIXED && anyHit);
    const uint32_t ida = __float_as_uint(p2.y), idb = __float_as_uint(q2.y);
    const bool ba = !isAny && ha && (ta < best || (ta == best && ida < bestId));
    const float b1 = ba ? ta : best;
    const uint32_t i1 = ba ? ida : bestId;
    const bool bb = !isAny && hb && (tb < b1 || (tb == b1 && idb < i1));
    best = bb ? tb : b1, bestId = bb ? idb : i1;
    if (MODE == TRAV_CLOSEST) {  // (pool mode recomputes u, v from the id: vertex_setup_ray)
      hit.u = bb ? ub : ba ? ua : hit.u, hit.v = bb ? vb : ba ? va : hit.v;
      hit.mesh = bb ? __float_as_uint(q2.z) : ba ? __float_as_uint(p2.z) : hit.mesh;
    }
    hit.t = best, hit.id = bestId;
    found = found || (isAny ? (ha || hb) : (ba || bb));
    if (MODE == TRAV_MIXED && shared && (ba || bb)) publish();
    return isAny && (ha || hb);
  }

  // Ray.cpp:9-24 on one 48-B record + the acceptance rule of RayTracer.h:40 (leaves with
  // more than two records: rt_options.bvh_leaf_max > 2).
  // Returns true when the ray is decided (any-hit rays only).
  template <bool STATS>
  RT_DEV bool test_record(const float4& q0, const float4& q1, const float4& q2, LaneStats& st) {
    if (STATS) st.tris++;
    float u, v, t;
    if (tri_test(o, d, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), u, v, t, DIVIDE) && t > 0.f) {
      if (MODE == TRAV_ANY || (MODE == TRAV_MIXED && anyHit)) {
        found = true;
        return true;
      }
      const uint32_t id = __float_as_uint(q2.y);
      if (t < best || (t == best && id < bestId)) {
        best = t, bestId = id, found = true;
        hit.t = t, hit.u = u, hit.v = v, hit.id = id, hit.mesh = __float_as_uint(q2.z);
        if (MODE == TRAV_MIXED && shared) publish();
      }
    }
    return false;
  }

  // pool mode: make this lane's closest hit visible to the other lanes / the pixel lane
  RT_DEV void publish() {
    const unsigned long long key = ((unsigned long long)__float_as_uint(hit.t) << 32) | hit.id;
    atomicMin(sharedKey + (int32_t)pj, key);
  }

  // pool mode: adopt a closer hit another lane has published for the same ray
  RT_DEV void refresh_best() {
    const unsigned long long key = sharedKey[(int32_t)pj];
    const float kt = __uint_as_float((uint32_t)(key >> 32));
    const uint32_t kid = (uint32_t)key;
    if (key != ~0ull && (kt < best || (kt == best && kid < bestId))) best = kt, bestId = kid;
  }
};

// RayTracer::rayTrace (RayTracer.h:27-53) through the flattened BVH.
//   ANY = false: closest positive t; on equal t the lowest global triangle id wins
//                (== the reference's strict '<' in mesh-then-triangle order)
//   ANY = true : stops at the first accepted triangle with t > 0 (Renderer.cpp:54
//                only uses the bool)
// `on` = this lane has a ray; lanes without one still take part in the wave loop.
template <bool ANY, bool STATS, int LT = LT_NONE>
RT_DEV bool traverse(const DevScene& S, bool on, f3 o, f3 d, uint32_t* stack, HitRec& hit, LaneStats& st) {
  Trav<ANY ? TRAV_ANY : TRAV_CLOSEST, LT> T;
  T.idle(stack);
  if (on) T.start(o, d, S.invBoxScale);
  PH(PH_SETUP);
  while (wave_ballot(T.live()) != 0) {
    T.template round<STATS>(S, st);
    PH(PH_PRIMARY);  // (round() itself books descent and leaf time; the primary cast is coherent)
  }
  if (!ANY && T.found) hit = T.hit;
  return T.found;
}

// The reference algorithm itself: every triangle, reference order (RayTracer.h:32-51).
template <bool ANY, bool STATS>
RT_DEV bool brute(const DevScene& S, f3 o, f3 d, HitRec& hit, LaneStats& st) {
  float best = 3.402823466e+38f;
  bool found = false;
  for (uint32_t i = 0; i < S.n_tris; i++) {
    const float4* r = S.trisRef + 3 * (size_t)i;  // wave-uniform address -> scalar loads
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    float u, v, t;
    if (tri_test(o, d, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), u, v, t, true)) {
      if (t > 0.f && t < best) {
        if (ANY) return true;
        best = t, found = true;
        hit.t = t, hit.u = u, hit.v = v, hit.id = i, hit.mesh = __float_as_uint(q2.z);
      }
    }
  }
  if (STATS) st.tris += S.n_tris;
  return found;
}

// `on`: lanes without a ray pass false (wave-uniform call sites, no early exits).
template <bool BRUTE, bool ANY, bool STATS, int LT = LT_NONE>
RT_DEV bool cast(const DevScene& S, bool on, f3 o, f3 d, uint32_t* stack, HitRec& hit, LaneStats& st) {
  if (BRUTE) return on && brute<ANY, STATS>(S, o, d, hit, st);
  return traverse<ANY, STATS, LT>(S, on, o, d, stack, hit, st);
}

// Renderer.cpp:274-277 dotArr: (w*a + u*b) + v*c per component
RT_DEV f3 interp3(const float* arr, uint4 tv, float w, float u, float v) {
  return w * ld(arr + 3 * (size_t)tv.x) + u * ld(arr + 3 * (size_t)tv.y) + v * ld(arr + 3 * (size_t)tv.z);
}

// ---------------------------------------------------------------- photon k-NN
// kdtree::knearest (kdtree.h:87-107 recursion, :180-195 driver) on the implicit
// median tree, restated step for step INCLUDING its quirks:
// heap seeded with array nodes [0,k), lagging m_bestdist, squared-vs-plain prune.
// std::make_heap/pop_heap/push_heap/sort_heap are restated from libstdc++
// (bits/stl_heap.h: __push_heap / __adjust_heap) because result ORDER feeds a
// float sum (Renderer.cpp:93-96).
struct Heap {
  uint2* e;  // [slot * BLOCK]: {distance bits, photon index} in one 8-byte word (one LDS op per move)
  RT_DEV float D(int s) const { return __uint_as_float(e[s * BLOCK].x); }
  RT_DEV uint32_t I(int s) const { return e[s * BLOCK].y; }
  RT_DEV void set(int s, float dv, uint32_t iv) const { e[s * BLOCK] = make_uint2(__float_as_uint(dv), iv); }
  RT_DEV void move(int dst, int src) const { e[dst * BLOCK] = e[src * BLOCK]; }

  RT_DEV void push_up(int hole, int top, float vd, uint32_t vi) const {
    int parent = (hole - 1) / 2;
    while (hole > top && D(parent) < vd) {
      move(hole, parent);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    set(hole, vd, vi);
  }
  RT_DEV void adjust(int hole, int len, float vd, uint32_t vi) const {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (D(child) < D(child - 1)) child--;
      move(hole, child);
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      move(hole, child - 1);
      hole = child - 1;
    }
    push_up(hole, top, vd, vi);
  }
  // std::pop_heap on [0,len): max goes to slot len-1, [0,len-1) stays a heap
  RT_DEV void pop(int len) const {
    if (len > 1) {
      const float vd = D(len - 1);
      const uint32_t vi = I(len - 1);
      move(len - 1, 0);
      adjust(0, len - 1, vd, vi);
    }
  }
  RT_DEV void make(int len) const {
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
      adjust(parent, len, D(parent), I(parent));
      if (parent == 0) return;
    }
  }
  RT_DEV void sort(int len) const {
    while (len > 1) {
      pop(len);
      len--;
    }
  }
};

RT_DEV float photon_dist(const DevScene& S, uint32_t n, f3 p, float4& pos) {
  pos = S.phPos[n];
  return dist3(mk(pos.x, pos.y, pos.z), p);
}

// kdtree::knearest (kdtree.h:87-107 recursion, :180-195 driver) over the tree's EXPLICIT topology (kd_build.hip
// k_kd_topology: child indices, axis and the parent's split in a record per photon).  One flat loop, one node per
// iteration, the same code for a lane that descends and a lane that comes back up:
//   * visit (kdtree.h:88-99): distance, insert if d < m_bestdist; stop below a zero m_bestdist (:101);
//   * descend into the near child (:102-104) and leave the far child on the lane's LDS stack — one word, its index;
//   * a lane with nowhere to descend pops the next far child and, in the next iteration, FIRST applies kdtree.h:105
//     (dx * dx >= m_bestdist, squared against plain, as there) with the m_bestdist of that moment and the split distance
//     rebuilt from the child's own record (parent's coordinate - query coordinate: the float subtraction the parent's
//     visit made) — a rejected child costs one iteration, nothing is unwound level by level.
// Round 3's walk carried the implicit tree's (b, e, depth, pending / side / parity bit masks) and spent a third of its
// vector instructions on range arithmetic, in two nested data-dependent loops whose trip counts the whole wave paid.
// Result-preserving extra prune, as before (546 -> ~1/6 of the node visits on C3): every photon of a far subtree lies
// beyond the split plane, so its float distance is >= |dx| * (1 - 1.5e-7) (monotone float subtraction; three roundings
// under the square root, one on it), and m_bestdist never increases for k >= 2 (the new value is the second largest of
// the old heap plus the new point): if |dx| * (1 - 4.8e-7) already reaches it, no node of that subtree can pass
// d < m_bestdist (kdtree.h:92) and the heap — hence the result — is the same whether or not the subtree is walked.
// (With k = 1 "the remaining k - 1" is empty and front() is the PREVIOUS insert, which may be larger than the one
// before: there only the reference's own test runs.)  The reference's own test is much weaker whenever m_bestdist < 1.
// After the call the heap holds the k results in ascending distance order.
// S16: the pending far children as 16-bit indices (maps of fewer than 65,535 photons: config 3's 35,744): half the stack's LDS,
// which with the k-heap is what caps the photon kernel's waves per CU.
constexpr uint32_t KD_NONE = 0x3fffffffu;
template <bool S16>
RT_DEV uint32_t knn_query(const DevScene& S, f3 p, int k, const Heap& H, uint32_t* stack) {
  using Entry = typename std::conditional<S16, uint16_t, uint32_t>::type;
  constexpr uint32_t NONE_E = S16 ? 0xffffu : KD_NONE;
  float4 pos;
  for (int j = 0; j < k; j++) H.set(j, photon_dist(S, (uint32_t)j, p, pos), (uint32_t)j);
  H.make(k);
  double bestdist = (double)H.D(0);
  float skip2 = (float)(bestdist * bestdist * (1.0 + 1e-6));
  uint32_t visited = 0;
  const bool mono = k >= 2;  // m_bestdist is non-increasing
  // (this lane's column: [row][lane] entries; 16-bit entries pack two rows into one of the stack's 256-byte rows)
  Entry* top = S16 ? reinterpret_cast<Entry*>(stack - (threadIdx.x & 63u)) + (threadIdx.x & 63u) : reinterpret_cast<Entry*>(stack);
  *top = (Entry)NONE_E;  // row 0: the sentinel an empty stack pops
  uint32_t cur = S.n_photons / 2u;  // the root: the median of [0, n)
  bool popped = false;
  while (cur != KD_NONE) {
    // the node's record: {position, left | axis << 30}{right, parent's split, parent's axis, -}: both halves in one line
    const uint4 ra = S.phTopo[2 * (size_t)cur], rb = S.phTopo[2 * (size_t)cur + 1];
    const float4 P = make_float4(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z), 0.f);
    const uint4 T = make_uint4(ra.w, rb.x, rb.y, rb.z);
    bool go = true;
    if (popped) {  // kdtree.h:105 for the activation that left this child behind, and the geometric prune
      const float pc = T.w == 0u ? p.x : T.w == 1u ? p.y : p.z;
      const double dxp = (double)(__uint_as_float(T.z) - pc);
      go = !(dxp * dxp >= bestdist) && !(mono && (dxp < 0 ? -dxp : dxp) * (1.0 - 4.8e-7) >= bestdist);
    }
    uint32_t next = KD_NONE;
    if (go) {
      ++visited;
      // kdtree.h:90-92 compares the fp32 distance sqrt(d2), widened, with m_bestdist.  The correctly rounded root is
      // only taken when d2 is not clearly out: d2 >= skip2 = m_bestdist^2 * (1 + 1e-6) rounded to float implies
      // sqrtf(d2) >= m_bestdist (a correctly rounded sqrt is monotone and within 6e-8 relative), i.e. no insert.
      const f3 dv = mk(P.x, P.y, P.z) - p;
      const float d2 = dot3(dv, dv);
      if (!(d2 >= skip2)) {
        const float dn = __builtin_sqrtf(d2);
        if ((double)dn < bestdist) {
          H.pop(k);                       // pop_heap
          bestdist = (double)H.D(0);      // front() of the remaining k-1, then pop_back
          H.set(k - 1, dn, cur);          // push_back(*root)
          H.push_up(k - 1, 0, dn, cur);   // push_heap
          skip2 = (float)(bestdist * bestdist * (1.0 + 1e-6));
        }
      }
      if (bestdist != 0) {
        const uint32_t axis = T.x >> 30;
        const float dx = axis == 0u ? dv.x : axis == 1u ? dv.y : dv.z;  // node coordinate - query coordinate (kdtree.h:102)
        const bool left = dx > 0.f;
        const uint32_t l = T.x & KD_NONE, r = T.y;
        const uint32_t far = left ? r : l;
        next = left ? l : r;
        // m_bestdist never increases (k >= 2), so a far side that both tests already reject now stays rejected: it is
        // not even recorded
        // (the entry is stored above the top unconditionally and claimed by a select, as Trav::round does: one row
        // beyond the deepest pending entry exists — the stack has a row per tree level plus the sentinel's)
        const double dxd = (double)dx;
        const bool out1 = dxd * dxd >= bestdist, out2 = (dxd < 0 ? -dxd : dxd) * (1.0 - 4.8e-7) >= bestdist;
        const bool record = (far != KD_NONE) & (!mono | !(out1 | out2));
        top[BLOCK] = (Entry)far;  // (KD_NONE truncates to the 16-bit sentinel)
        top += record ? BLOCK : 0;
      }
    }
    popped = next == KD_NONE;
    if (popped) {
      const uint32_t e = *top;
      next = e == NONE_E ? KD_NONE : e;
      if (e != NONE_E) top -= BLOCK;  // (the sentinel stays)
    }
    cur = next;
  }
  H.sort(k);
  return visited;
}

// ---------------------------------------------------------------- shading
constexpr int POOL_L = 3;                              // lights handled by the shadow-ray pool

RT_DEV uint32_t lanes_below(uint64_t m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Renderer.cpp:42-43: interpolated shading normal and hit point
template <bool FAST = false>
RT_DEV void vertex_setup(const DevScene& S, const HitRec& h, f3& hitNormal, f3& point) {
  const float w = 1.f - h.u - h.v;
  const uint4 tv = S.triShade[h.id];
  hitNormal = unit3<FAST>(interp3(S.vnrm, tv, w, h.u, h.v));
  point = interp3(S.vpos, tv, w, h.u, h.v);
}

// The same from the triangle id alone (vertex pool: a bounce hit comes back as a
// (t, id) key).  u, v are recomputed by Ray.cpp:9-24 on exactly the operands the
// walker's record holds — p0 and the float edge differences bvh_build.cpp stores —
// so they are the bits the walker had; the vertex loads are the ones the
// interpolation needs anyway.
template <bool FAST = false>
RT_DEV void vertex_setup_ray(const DevScene& S, uint32_t id, f3 o, f3 d, f3& hitNormal, f3& point, uint32_t& mesh) {
  const uint4 tv = S.triShade[id];
  const f3 p0 = ld(S.vpos + 3 * (size_t)tv.x), p1 = ld(S.vpos + 3 * (size_t)tv.y), p2 = ld(S.vpos + 3 * (size_t)tv.z);
  float u, v, t;
  tri_test(o, d, p0, p1 - p0, p2 - p0, u, v, t, true);  // (once per vertex: the division; the same bits as the walker's)
  const float w = 1.f - u - v;
  hitNormal = unit3<FAST>(interp3(S.vnrm, tv, w, u, v));
  point = w * p0 + u * p1 + v * p2;
  mesh = tv.w;
}

// Renderer.cpp:63-104: photon-map radiance estimate at the vertex
template <bool STATS>
RT_DEV f3 shade_photon(const DevScene& S, const RenderArgs& A, f3 rayDir, const HitRec& h, const Lds& L, f3 hitNormal,
                       f3 point, LaneStats& st) {
  const rt_material mat = S.mats[h.mesh];
  const Heap H{L.heap};
  const int k = (int)A.k;
  st.knn++;
  const uint32_t vis = A.kd16 ? knn_query<true>(S, point, k, H, L.stack) : knn_query<false>(S, point, k, H, L.stack);
  if (STATS) st.kd += vis;
  const float4 far = S.phPos[H.I(k - 1)];
  const float r = dist3(mk(far.x, far.y, far.z), point);
  const float area = (float)(3.14159265358979323846 * (double)r * (double)r);
  f3 avg = mk(0.f, 0.f, 0.f), radiance = mk(0.f, 0.f, 0.f);
  for (int j = 0; j < k; j++) {
    avg = avg + f4xyz(S.phDir[H.I(j)]);
    radiance = radiance + mk(1.f, 1.f, 1.f);
  }
  radiance = radiance / area;
  radiance = radiance / (float)(int)A.photons_requested;
  radiance = radiance * 100.f;  // Renderer.h:45
  const f3 bsdf = bsdf_eval(mat, hitNormal, unit3(avg), -rayDir);
  return mk(0.f, 0.f, 0.f) + radiance * bsdf;
}

// Renderer.cpp:49-60, one light after the other (brute-force variant and scenes
// with more than POOL_L lights).  Called by lanes that own a vertex only.
template <bool BRUTE, bool STATS>
RT_DEV f3 shade_direct_seq(const DevScene& S, Rng& g, f3 rayDir, const HitRec& h, uint32_t* stack, f3 hitNormal,
                           f3 point, LaneStats& st) {
  const BsdfBase base = bsdf_base(S.matsDev[h.mesh], hitNormal, -rayDir);  // the light-independent half, once
  f3 color = mk(0.f, 0.f, 0.f);
  for (uint32_t li = 0; li < S.n_lights; li++) {
    const rt_light Lt = S.lights[li];
    const f3 toLight = light_sample(g, Lt) - point;
    HitRec tmp;
    st.shadow++;
    if (cast<BRUTE, true, STATS>(S, true, point, toLight, stack, tmp, st)) continue;
    const f3 bsdf = bsdf_apply(base, toLight);
    const f3 radiance = light_eval(Lt, point);
    color = color + radiance * bsdf;
  }
  return color;
}

// ---------------------------------------------------------------- vertex pool
// Everything a path vertex sends out is known as soon as the vertex exists: the
// n_lights shadow rays (Renderer.cpp:52-54: the light samples are drawn
// unconditionally, in light order) AND the bounce ray (Renderer.cpp:164-166: the
// hemisphere sample depends on the normal and the stream only).  So all
// (n_lights + 1) x (lanes with a vertex) rays of a wave go into one LDS pool and the
// 64 lanes work through it as WORKERS — a lane whose ray is decided fetches the next
// undecided one (ballot + mbcnt compaction, no atomics) instead of idling until
// the slowest of 64 rays ends.  Per pixel lane the pool holds: vertex position (3
// words), one direction per light (3 x POOL_L), the bounce direction (3), the bounce
// result as a 64-bit key (t, id).  Shadow results are one bit per (light, lane).  Arithmetic and draw order per pixel are exactly those of
// the sequential code, so results are bit-identical.
constexpr int VP_PT = 0, VP_DIR = 192, VP_BDIR = VP_DIR + 192 * POOL_L, VP_KEY = VP_BDIR + 192, VP_LIST = VP_KEY + 128,
              VP_RES = VP_LIST + 32;
constexpr int VP_WORDS = VP_RES + 2 * POOL_L + 2;
constexpr int VP_COMPACT_SAVES = 64 * POOL_L;  // words a compact pool is shorter by
// offsets of everything behind the per-light block, by layout
template <int CP> struct VpLayout {  // CP = 0: full, 1: light parameters, 2: ... and no bounce directions, 3: keys, list and result bits only
  static constexpr int PER_LIGHT = CP ? 128 : 192;
  static constexpr int BDIR = CP == 3 ? 0 : VP_DIR + PER_LIGHT * POOL_L, KEY = BDIR + (CP >= 2 ? 0 : 192), LIST = KEY + 128, RES = LIST + 32;
  static constexpr int WORDS = RES + 2 * POOL_L + 2;
};
static_assert(VpLayout<3>::WORDS == (int)rtbvh::kWavePoolWordsMin && VpLayout<3>::KEY % 2 == 0, "the smallest pool (bvh_build.h sizes the depth cap with it)");
static_assert(VpLayout<0>::WORDS == VP_WORDS && VpLayout<1>::WORDS == VP_WORDS - VP_COMPACT_SAVES &&
              VpLayout<2>::WORDS == VP_WORDS - 2 * VP_COMPACT_SAVES, "pool layouts");
static_assert(VpLayout<1>::KEY % 2 == 0 && VpLayout<2>::KEY % 2 == 0, "64-bit keys need 8-byte alignment");
static_assert(VP_KEY % 2 == 0, "64-bit keys need 8-byte alignment");
static_assert(VP_WORDS == (int)rtbvh::kWavePoolWords && BLOCK == (int)rtbvh::kStackRowWords, "bvh_build.h sizes the depth cap with these");

template <bool STATS, int LT>
RT_DEV f3 vertex_pool(const DevScene& S, bool alive, bool bounce, Rng& g, f3 rayDir, uint32_t mesh, f3 hitNormal,
                      f3& point, f3& bdir, uint32_t* stack, uint32_t* pool, HitRec& next, bool& nextFound, LaneStats& st) {
  constexpr bool FR = (LT & LT_FASTDET) != 0;  // 1 / length by rtd::recip_fast (see rtd::unit3)
  const uint32_t lane = threadIdx.x & 63u, nl = S.n_lights;
  constexpr bool CP3 = (LT & LT_COMPACT3) != 0, CP2 = CP3 || (LT & LT_COMPACT2) != 0, CP = CP2 || (LT & LT_COMPACT) != 0;
  using VP = VpLayout<CP3 ? 3 : CP2 ? 2 : CP ? 1 : 0>;
  float* fp = reinterpret_cast<float*>(pool);
  // 32 words: rank -> pixel lane (64 bytes) while rays are handed out; then rank -> victim
  // word while stealing (min(victims, free lanes) <= 32 entries)
  uint32_t* list = pool + VP::LIST;
  uint8_t* listB = reinterpret_cast<uint8_t*>(list);
  uint32_t* res = pool + VP::RES;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(pool + VP::KEY);
  const uint64_t amask = wave_ballot(alive);
  const uint32_t n = (uint32_t)__popcll(amask);
  f3 color = mk(0.f, 0.f, 0.f);
  nextFound = false;
  uint32_t gsave = 0;  // (LT_COMPACT3) this lane's engine state before its light draws
  if (n == 0) return color;
  PH(PH_VSETUP);
  PHC(PH_N_POOLS);
  if (alive) {
    if (!CP3) fp[VP_PT + lane] = point.x, fp[VP_PT + 64 + lane] = point.y, fp[VP_PT + 128 + lane] = point.z;
    // draw order of the reference: the light samples in light order (Renderer.cpp:52),
    // then the hemisphere sample (Renderer.cpp:164)
    if (CP3) gsave = g.s;  // (the light samples are re-drawn from here wherever they are needed)
    for (uint32_t l = 0; l < nl; l++) {
      if (CP3) {
        g.next(), g.next();  // (light_sample_params' two engine calls)
      } else if (CP) {
        float rh, rv;
        light_sample_params(g, S.lights[l], rh, rv);
        fp[VP_DIR + (2 * l + 0) * 64 + lane] = rh, fp[VP_DIR + (2 * l + 1) * 64 + lane] = rv;
      } else {
        const f3 tl = light_sample(g, S.lights[l]) - point;
        fp[VP_DIR + (3 * l + 0) * 64 + lane] = tl.x, fp[VP_DIR + (3 * l + 1) * 64 + lane] = tl.y, fp[VP_DIR + (3 * l + 2) * 64 + lane] = tl.z;
      }
    }
    if (bounce) {
      // (the caller's copy is read back from the pool after the loop: `point` and the bounce
      // direction do not occupy registers while the wave traverses)
      const f3 bd = hemisphere_sample<FR>(g, hitNormal);
      if (CP2) bdir = bd;  // (stays in the caller's registers)
      else fp[VP::BDIR + lane] = bd.x, fp[VP::BDIR + 64 + lane] = bd.y, fp[VP::BDIR + 128 + lane] = bd.z;
      keys[lane] = ~0ull;
    }
    listB[lanes_below(amask)] = (uint8_t)lane;
    st.shadow += nl;
    if (bounce) st.closest++;
  }
  if (lane < 2 * POOL_L) res[lane] = 0;
  wave_sync();
  PH(PH_FILL);
  const uint32_t kinds = nl + (bounce ? 1u : 0u), R = n * kinds;
  uint32_t head = 0, myK = 0, myJ = 0;
  Trav<TRAV_MIXED, LT> T;
  T.idle(stack);
  T.sharedKey = keys, T.pj = 0;
  uint32_t* stackBase = stack - lane;
  if (!(LT & LT_NOPRIO)) __builtin_amdgcn_s_setprio(1);
  for (;;) {
    const uint64_t idle = wave_ballot(!T.live());
    const int nIdle = __popcll(idle);
    // a lane may be given a ray (hand-out) or a subtree of one (stealing); the ray is
    // started in ONE place below, so that the traversal state has a single definition
    // point per iteration (two start sites cost ~25 register copies per round)
    bool newRay = false, newShared = false;
    uint32_t newK = 0, newJ = 0;
    int32_t newNode = (LT & LT_Q8) ? (int32_t)rtbvh::kQ8RootOffset : 0;  // (the root's record)
    f3 dv = mk(0.f, 0.f, 0.f);  // (CP2) direction of the ray a thief joins, from the victim's registers
    if (CP2 && bounce && head == 0) {
      // first hand-out (every lane is free): each bounce ray to its own pixel lane, the first
      // shadow rays to the lanes without a vertex — the same 64 ranks as below, dealt differently
      if (alive) {
        newRay = true, newK = nl, newJ = lane;
      } else {
        const uint32_t r = n + lanes_below(~amask);
        if (r < R) {
          const uint32_t k = (r >= n) + (r >= 2 * n) + (r >= 3 * n);  // >= 1 here
          newRay = true, newK = k - 1u, newJ = listB[r - k * n];
        }
      }
      head = 64;
    } else if (head < R) {
      if (nIdle >= (int)S.refillT || nIdle == 64) {
        const uint32_t r = head + lanes_below(idle);
        if (!T.live() && r < R) {
          // the bounce rays (closest hit: the longest walks) are handed out first
          uint32_t k = (r >= n) + (r >= 2 * n) + (r >= 3 * n);  // r / n for kinds <= POOL_L + 1
          const uint32_t j = listB[r - k * n];
          k = bounce ? (k == 0 ? nl : k - 1) : k;
          newRay = true, newK = k, newJ = j;
        }
        head += (uint32_t)nIdle;
      }
      PH(PH_HANDOUT);
    } else if (nIdle >= (int)S.stealT) {
      // The pool is empty and many workers are free: the tail.  A few long rays would
      // now keep 64 lanes waiting (measured: 75 % of the rounds ran with 8 live lanes).
      // Free lanes take over the BOTTOM pending subtree of a busy lane's stack (the
      // farthest, usually largest one) and walk it for the same ray.  The victim's slot
      // is overwritten with TERM, which is what it must pop there anyway; any-hit rays
      // share one result bit, closest-hit rays one atomic-min key, so the outcome is
      // the one a single walker produces.
      // One steal event hands out up to four bottom entries per busy lane (pass p takes
      // every victim's next one), so that a single long ray with a deep stack refills the
      // wave at once instead of doubling its walkers round by round.  list16[slot] =
      // victim lane | entry index << 6; the thief fetches (kind, pixel lane) from the
      // victim's registers.
      uint16_t* list16 = reinterpret_cast<uint16_t*>(list);
      uint32_t given = 0;
      // entries this lane could hand over
      const int avail = T.live() ? T.depth() - T.stolen : 0;
      // (the passes only write the list; the lane's own state changes once, after them —
      // per-lane flags updated inside a loop with exits cost a dozen mask instructions per pass)
      int gave = 0;
      for (int pass = 0; pass < 4; pass++) {
        const bool canGive = avail > pass;
        const uint64_t vmask = wave_ballot(canGive);
        if (vmask == 0 || given >= (uint32_t)nIdle) break;
        const uint32_t slot = given + lanes_below(vmask);
        const bool gives = canGive && slot < (uint32_t)nIdle;
        if (gives) list16[slot] = (uint16_t)(lane | ((uint32_t)(T.stolen + pass) << 6));
        gave += gives ? 1 : 0;
        given += (uint32_t)__popcll(vmask);
      }
      if (gave) {
        if (!T.shared && !T.anyHit && T.found) T.publish();  // what it has found so far
        T.shared = true;
        T.stolen += gave;
      }
      if (given != 0) {
        given = given < (uint32_t)nIdle ? given : (uint32_t)nIdle;
        wave_sync();
        const uint32_t q = lanes_below(idle);
        const bool thief = !T.live() && q < given;
        const uint32_t w = thief ? (uint32_t)list16[q] : lane;
        const uint32_t v = w & 63u, e = w >> 6;
        const uint32_t kj = (uint32_t)__shfl((int)(myK | (myJ << 8)), (int)v, 64);  // (all lanes take part)
        if (CP2) dv = mk(__shfl(T.d.x, (int)v, 64), __shfl(T.d.y, (int)v, 64), __shfl(T.d.z, (int)v, 64));
        if (thief) {
          uint32_t* slot = stackBase + (e + 1u) * BLOCK + v;  // (row 0 is the sentinel)
          newNode = (int32_t)*slot;
          *slot = (uint32_t)TERM;
          newRay = true, newShared = true, newK = kj & 255u, newJ = kj >> 8;
        }
        wave_sync();
      }
      PH(PH_STEAL);
    }
    const uint32_t gsj = CP3 ? (uint32_t)__shfl((int)gsave, (int)newJ, 64) : 0u;  // (all lanes take part)
    const f3 pj3 = CP3 ? mk(__shfl(point.x, (int)newJ, 64), __shfl(point.y, (int)newJ, 64), __shfl(point.z, (int)newJ, 64)) : mk(0.f, 0.f, 0.f);
    if (newRay) {
      const f3 pj = CP3 ? pj3 : mk(fp[VP_PT + newJ], fp[VP_PT + 64 + newJ], fp[VP_PT + 128 + newJ]);
      f3 dj;
      if (CP) {
        if (newK < nl) {  // rebuild the direction from the light sample's two parameters
          const rt_light& Lt = S.lights[newK];
          float rh, rv;
          if (CP3) {  // ... which are re-drawn: the pixel lane's engine state before its light draws, 2 calls per earlier light
            Rng t{gsj};
            for (uint32_t q = 0; q < 2u * newK; ++q) t.next();
            light_sample_params(t, Lt, rh, rv);
          } else {
            rh = fp[VP_DIR + (2 * newK + 0) * 64 + newJ], rv = fp[VP_DIR + (2 * newK + 1) * 64 + newJ];
          }
          dj = light_point(Lt, rh, rv) - pj;
        } else if (CP2) {
          dj = newShared ? dv : bdir;  // a bounce ray: stolen (the victim's direction) or this lane's own
        } else {
          dj = mk(fp[VP::BDIR + newJ], fp[VP::BDIR + 64 + newJ], fp[VP::BDIR + 128 + newJ]);
        }
      } else {
        const uint32_t src = newK < nl ? VP_DIR + 192 * newK : VP::BDIR;
        dj = mk(fp[src + newJ], fp[src + 64 + newJ], fp[src + 128 + newJ]);
      }
      T.start(pj, dj, S.invBoxScale);
      if (T.live()) T.cur = newNode;  // (a NaN ray stays dead)
      T.anyHit = newK < nl, T.pj = newJ, T.shared = newShared;
      myK = newK, myJ = newJ;
    }
    if (wave_ballot(T.live()) == 0) break;
    // Rounds until enough lanes are free for the next hand-out or steal (the same
    // thresholds as above, so the schedule is the one a check per round gives — without
    // walking through the bookkeeping in the rounds where it cannot do anything)
    const int need = min(64, head < R ? (int)S.refillT : (int)S.stealT);
    for (;;) {
      if (T.shared && T.live()) {
        // several lanes may now serve this ray: stop at a decided any-hit
        // ray, prune with the closest hit anyone has found so far
        if (T.anyHit) {
          if ((res[myK * 2 + (myJ >> 5)] >> (myJ & 31)) & 1u) T.cur = TERM;
        } else {
          T.refresh_best();
        }
      }
      const bool was = T.live();
      PH(PH_POOLMISC);
      if (head >= R) PHC(PH_TAIL);
      T.template round<STATS>(S, st);
      if (was && !T.live() && T.found) {
        if (myK < nl) atomicOr(&res[myK * 2 + (myJ >> 5)], 1u << (myJ & 31));
        else if (!T.shared) T.publish();  // shared rays publish every improvement as it happens
      }
      if (__popcll(wave_ballot(!T.live())) >= need) break;
    }
  }
  if (!(LT & LT_NOPRIO)) __builtin_amdgcn_s_setprio(0);
  wave_sync();
  PH(PH_POOLMISC);
  // `point` and the bounce direction come back from the pool (the same bits; lanes without a
  // vertex read stale words nobody uses): nothing of them is live while the wave traverses
  const f3 pt = CP3 ? point : mk(fp[VP_PT + lane], fp[VP_PT + 64 + lane], fp[VP_PT + 128 + lane]);
  point = pt;
  if (bounce && !CP2) bdir = mk(fp[VP::BDIR + lane], fp[VP::BDIR + 64 + lane], fp[VP::BDIR + 128 + lane]);
  if (alive) {
    const BsdfBase base = bsdf_base<FR>(S.matsDev[mesh], hitNormal, -rayDir);  // the light-independent half, once
    Rng relight{gsave};
    for (uint32_t l = 0; l < nl; l++) {
      float rh3 = 0.f, rv3 = 0.f;
      if (CP3) light_sample_params(relight, S.lights[l], rh3, rv3);  // (every light: the stream moves on whether it is occluded or not)
      if ((res[l * 2 + (lane >> 5)] >> (lane & 31)) & 1u) continue;  // occluded (Renderer.cpp:54-55)
      const f3 toLight = CP3 ? light_point(S.lights[l], rh3, rv3) - pt
                         : CP ? light_point(S.lights[l], fp[VP_DIR + (2 * l + 0) * 64 + lane], fp[VP_DIR + (2 * l + 1) * 64 + lane]) - pt
                              : mk(fp[VP_DIR + (3 * l + 0) * 64 + lane], fp[VP_DIR + (3 * l + 1) * 64 + lane], fp[VP_DIR + (3 * l + 2) * 64 + lane]);
      const f3 bsdf = bsdf_apply<FR>(base, toLight);
      const f3 radiance = light_eval(S.lights[l], pt);
      color = color + radiance * bsdf;
    }
    if (bounce) {
      const unsigned long long key = keys[lane];
      nextFound = key != ~0ull;
      next.t = __uint_as_float((uint32_t)(key >> 32)), next.id = (uint32_t)key;
      next.u = next.v = 0.f, next.mesh = 0u;  // vertex_setup_ray works from the id
    }
  }
  wave_sync();  // the pool is rewritten by the next vertex
  PH(PH_BSDF);
  return color;
}

RT_DEV uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

RT_DEV void flush_stats(const LaneStats& st, unsigned long long* counters, bool stats) {
  const uint32_t c = wave_sum(st.closest), s = wave_sum(st.shadow), q = wave_sum(st.knn);
  uint32_t n = 0, t = 0, kd = 0;
  uint32_t wn = 0, wl = 0;
  uint32_t lw = 0, li = 0, fr = 0;
  if (stats) fr = wave_sum(st.frames), n = wave_sum(st.nodes), t = wave_sum(st.tris), kd = wave_sum(st.kd), wn = wave_sum(st.wnode), wl = wave_sum(st.wleaf), lw = wave_sum(st.lwait), li = wave_sum(st.lidle);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counters[RTK_CNT_CLOSEST], (unsigned long long)c);
    atomicAdd(&counters[RTK_CNT_SHADOW], (unsigned long long)s);
    if (q) atomicAdd(&counters[RTK_CNT_KNN], (unsigned long long)q);
    if (stats) {
      atomicAdd(&counters[RTK_CNT_NODES], (unsigned long long)n);
      atomicAdd(&counters[RTK_CNT_TRIS], (unsigned long long)t);
      if (kd) atomicAdd(&counters[RTK_CNT_KD], (unsigned long long)kd);
      atomicAdd(&counters[RTK_CNT_WNODE], (unsigned long long)wn);
      atomicAdd(&counters[RTK_CNT_WLEAF], (unsigned long long)wl);
      atomicAdd(&counters[RTK_CNT_LWAIT], (unsigned long long)lw);
      atomicAdd(&counters[RTK_CNT_LIDLE], (unsigned long long)li);
      if (fr) atomicAdd(&counters[RTK_CNT_FRAMES], (unsigned long long)fr);
    }
  }
}

// The same for the one-wave-per-workgroup kernels, whose frames have up to hundreds of thousands of waves: the three
// ray / query counts go to 1,024 striped slots behind the counter block (counters + RTK_CNT_COUNT: [1024][4]) and
// k_fold_stripes adds them up after the kernel — 262 k waves adding to ONE address each spent 4 of the photon frame's
// 17 ms queueing at the memory side (same-address atomics serialise there).
RT_DEV void flush_stats_striped(const LaneStats& st, unsigned long long* counters) {
  const uint32_t c = wave_sum(st.closest), s = wave_sum(st.shadow), q = wave_sum(st.knn);
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* stripe = counters + RTK_CNT_COUNT + 4u * (blockIdx.x & 1023u);
    if (c) atomicAdd(&stripe[0], (unsigned long long)c);
    if (s) atomicAdd(&stripe[1], (unsigned long long)s);
    if (q) atomicAdd(&stripe[2], (unsigned long long)q);
  }
}
__global__ __launch_bounds__(1024) void k_fold_stripes(unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long sh[3][16];
  unsigned long long* stripe = counters + RTK_CNT_COUNT + 4u * threadIdx.x;
  unsigned long long v[3] = {stripe[0], stripe[1], stripe[2]};
  stripe[0] = stripe[1] = stripe[2] = 0;  // ready for the next launch
  for (int j = 0; j < 3; ++j) {
    for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
    if ((threadIdx.x & 63u) == 0) sh[j][threadIdx.x >> 6] = v[j];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; i++) t += sh[threadIdx.x][i];
    const int which = threadIdx.x == 0 ? RTK_CNT_CLOSEST : threadIdx.x == 1 ? RTK_CNT_SHADOW : RTK_CNT_KNN;
    if (t) atomicAdd(&counters[which], t);
  }
}

// [levels][64] stack, then (photon variants) [k][64] heap distances and [k][64] heap ids
template <bool PHOTON>
RT_DEV Lds carve_lds(uint32_t* base, uint32_t levels = STACK, uint32_t kslots = KMAX) {
  Lds L;
  L.stack = base + (threadIdx.x & 63u);
  L.heap = PHOTON ? reinterpret_cast<uint2*>(base + levels * BLOCK) + (threadIdx.x & 63u) : nullptr;
  return L;
}

// ---------------------------------------------------------------- integrate
// Renderer::render's per-sample body (Renderer.cpp:227-258) + calculateColorRay /
// calculateColorPath (:106-201) with the recursion unrolled to a loop.  Control
// flow is wave-uniform (per-lane `alive` flags instead of early exits) because the
// direct-lighting step exchanges rays between lanes through LDS.
// One wave tile: lane = (pixel pl of the tile, sample slot sj).  The wave integrates
// S = 1 << sshift consecutive samples of P = 64 >> sshift pixels side by side.

template <bool BRUTE, bool PHOTON, bool POOLED, bool STATS, int LT>
RT_DEV void render_tile(const DevScene& S, const RenderArgs& A, float4* __restrict__ accum, const Lds& L, uint32_t* pool,
                        float* ex, uint32_t wave, LaneStats& st) {
  constexpr bool FR = (LT & LT_FASTDET) != 0;  // 1 / length by rtd::recip_fast (see rtd::unit3)
  const uint32_t lane = threadIdx.x & 63u;
  // lane = (pixel pl of the wave tile, sample slot sj): the wave integrates
  // S = 1 << sshift consecutive samples of P = 64 >> sshift pixels side by side
  const uint32_t tile = A.tiles[wave];
  const uint32_t S_ = 1u << A.sshift, P_ = 64u >> A.sshift;
  const uint32_t pl = lane & (P_ - 1u), sj = lane >> (6u - A.sshift);
  const uint32_t wsh = (uint32_t)__builtin_ctz(A.tileW);
  const uint32_t px = (tile & 0xffffu) + (pl & (A.tileW - 1u)), py = (tile >> 16) + (pl >> wsh);
  const bool inImage = px < A.width && py < A.height;
  const bool owner = inImage && sj == 0;  // adds this pixel's samples, in order
  const uint32_t pix = inImage ? py * A.width + px : 0u;
  float4 sum = owner ? accum[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr bool pooled = POOLED;  // chosen by the launcher: BVH, direct lighting, n_lights <= POOL_L
  const int nvert = A.mode == RT_MODE_PATH ? (int)A.max_depth : 1;
  for (uint32_t base = A.s0; base < A.s1; base += S_) {
    const uint32_t i = base + sj;
    const bool active = inImage && i < A.s1;
    Rng g{rt_stream_seed(A.seed, RT_STREAM_PIXEL, pix, i)};
    float sx, sy;
    jitter_sample(g, (int)i, (int)A.spp, sx, sy);
    f3 o, d;
    camera_ray<FR>(S.cam, ((float)px + sx) / (float)A.width, 1.f - ((float)py + sy) / (float)A.height, o, d);
    f3 c0 = mk(0.f, 0.f, 0.f), c1 = c0, c2 = c0;
    bool primary = true, alive = active;
    if constexpr (pooled) {
      // primary ray (coherent: traced in lock step), then one pool per vertex
      HitRec h;
      if (alive) st.closest++;
      const bool hit0 = cast<false, false, STATS, LT>(S, alive, o, d, L.stack, h, st);
      if (alive && !hit0) primary = false, alive = false;
      for (int depth = 0; depth < nvert; depth++) {
        if (wave_ballot(alive) == 0) break;
        const bool bounce = A.mode == RT_MODE_PATH && depth + 1 < nvert;  // wave-uniform
        f3 nrm = mk(0.f, 0.f, 0.f), pt = nrm, bdir = nrm;
        uint32_t mesh = 0;
        // (the vertex set-up is three dependent loads deep: it goes out ahead of other waves'
        // arithmetic, like the pool loop that follows; +0.4 % on C2)
        if (!(LT & LT_NOPRIO)) __builtin_amdgcn_s_setprio(1);
        if (alive) vertex_setup_ray<FR>(S, h.id, o, d, nrm, pt, mesh);
        // (Renderer.cpp:164: the hemisphere sample is drawn after every shaded vertex;
        // after the LAST one the reference draws it too but never traces it, and the
        // stream ends there — the pool only draws it when a bounce ray follows)
        HitRec nh;
        bool nfound;
        const f3 c = vertex_pool<STATS, LT>(S, alive, bounce, g, d, mesh, nrm, pt, bdir, L.stack, pool, nh, nfound, st);
        if (alive) {
          if (depth == 0) c0 = c;
          else if (depth == 1) c1 = c;
          else c2 = c;
          o = pt, d = bdir, h = nh;
          if (!nfound) alive = false;
        }
      }
    } else
    for (int depth = 0; depth < nvert; depth++) {
      HitRec h;
      if (alive) st.closest++;
      const bool hitv = cast<BRUTE, false, STATS>(S, alive, o, d, L.stack, h, st);
      if (alive && !hitv) {
        if (depth == 0) primary = false;
        alive = false;
      }
      if (wave_ballot(alive) == 0) break;
      f3 nrm = mk(0.f, 0.f, 0.f), pt = nrm, c = nrm;
      if (alive) vertex_setup<FR>(S, h, nrm, pt);
      if (PHOTON) {
        if (alive) c = shade_photon<STATS>(S, A, d, h, L, nrm, pt, st);
      } else {
        if (alive) c = shade_direct_seq<BRUTE, STATS>(S, g, d, h, L.stack, nrm, pt, st);
      }
      if (alive) {
        if (depth == 0) c0 = c;
        else if (depth == 1) c1 = c;
        else c2 = c;
      }
      if (A.mode != RT_MODE_PATH) break;
      if (alive) {
        d = hemisphere_sample(g, nrm);  // drawn after every shaded vertex (Renderer.cpp:164)
        o = pt;
      }
    }
    // calculateColorPath returns c0 + (c1 + (c2 + 0)) for finalDepth <= 3
    PH(PH_VSETUP);
    const f3 total = c0 + (c1 + (c2 + mk(0.f, 0.f, 0.f)));
    const float r0 = clamp01(total.x), r1 = clamp01(total.y), r2 = clamp01(total.z);
    if (A.sshift == 0) {
      if (active) {
        sum.x += r0, sum.y += r1, sum.z += r2;
        if (primary) sum.w += 1.f;
      }
    } else {
      // hand the sample to the pixel's owner lane, which adds samples base.. in order
      // (Renderer.cpp:258: updateImage += colorResponse, i = 0..N-1)
      ex[lane] = r0, ex[64 + lane] = r1, ex[128 + lane] = r2, ex[192 + lane] = primary ? 1.f : 0.f;
      wave_sync();
      if (owner) {
        const uint32_t cnt = A.s1 - base < S_ ? A.s1 - base : S_;
        for (uint32_t jj = 0; jj < cnt; jj++) {
          const uint32_t q = jj * P_ + pl;
          sum.x += ex[q], sum.y += ex[64 + q], sum.z += ex[128 + q];
          sum.w += ex[192 + q];  // adds 1.0 or an exact 0.0
        }
      }
      wave_sync();
    }
    PH(PH_ACCUM);
  }
  if (owner) accum[pix] = sum;
}

template <bool BRUTE, bool PHOTON, bool POOLED, bool STATS, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void k_render(DevScene S, RenderArgs A, float4* __restrict__ accum,
                                                  unsigned long long* __restrict__ counters) {
  static_assert(BLOCK == 64, "one wave per workgroup");
  static_assert(!POOLED || (!BRUTE && !PHOTON), "the vertex pool serves BVH direct lighting");
  // dynamic LDS (render_lds_bytes): [levels][64] traversal stack, sized from the
  // depth of THIS scene's trees so that LDS does not cap occupancy; then the photon
  // k-heap or the shadow-ray pool
  uint32_t* lds = g_lds;
  const Lds L = carve_lds<PHOTON>(lds, A.stackLevels, A.k);
  uint32_t* pool = lds + (A.stackLevels + (PHOTON ? 2 * A.k : 0)) * BLOCK;
  // [4][64] sample results: share the pool's words, or (no pool) the first four stack
  // levels — both idle when a sample is handed over
  float* ex = reinterpret_cast<float*>(POOLED ? pool : lds);
  LaneStats st;
  // A.tilesPerBlock consecutive wave tiles per workgroup (1 unless RT_TILES_PER_BLOCK says otherwise: more of them
  // unbalance the grid — C3 13.7 / 14.8 / 17.8 ms at 4 / 8 / 16)
  const uint32_t t0 = blockIdx.x * A.tilesPerBlock, t1 = min(A.n_tiles, t0 + A.tilesPerBlock);
  for (uint32_t t = t0; t < t1; ++t) render_tile<BRUTE, PHOTON, POOLED, STATS, LT_NONE>(S, A, accum, L, pool, ex, t, st);
  if (STATS) flush_stats(st, counters, true);
  else flush_stats_striped(st, counters);  // (launch_render2 folds the stripes)
}

// The pooled integrator as PERSISTENT workgroups: one workgroup of up to 16 waves per CU
// for the whole launch.  Its LDS holds ONE copy of the top of the BVH (node records
// [0, topK), breadth-first) in front of the waves' private regions (traversal stack +
// vertex pool), and every wave draws wave tiles from a global counter until none is
// left — so a CU never idles behind a slow neighbour wave, and the node fetches of the
// upper tree levels (all of them for the 1.2k-triangle scene) never touch the vector L1.
// Waves never synchronise with each other after the tree copy.
template <bool STATS, int LT>
RT_DEV void persist_body(const DevScene& S, const RenderArgs& A, float4* __restrict__ accum, unsigned long long* __restrict__ counters) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  {
    uint4* dst = reinterpret_cast<uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < 2u * S.topK; i += blockDim.x) dst[i] = S.nodes[i];
  }
  __syncthreads();  // the only workgroup-wide barrier
  uint32_t* mine = g_lds + 8u * S.topK + wv * A.waveWords;
  Lds L = carve_lds<false>(mine, A.stackLevels, 0);
  uint32_t* pool = mine + A.stackLevels * BLOCK;
  // (the 256-word exchange area of the sample adds: in the pool — or, where the pool is smaller than that, in the stack rows,
  // which are idle between two vertices; Trav::start rewrites the sentinel row)
  float* ex = reinterpret_cast<float*>((LT & LT_COMPACT3) ? mine : pool);
  LaneStats st;
#ifdef RT_PHASE_TIMING
  if (threadIdx.x < 24) g_phAcc[threadIdx.x] = 0;
  if (threadIdx.x == 0) g_phT0 = __builtin_amdgcn_s_memtime();
  __syncthreads();
#endif
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(A.tileCounter, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= A.n_tiles) break;
    render_tile<false, false, true, STATS, LT>(S, A, accum, L, pool, ex, t, st);
  }
#ifdef RT_PHASE_TIMING
  __syncthreads();
  if (threadIdx.x < PH_COUNT) atomicAdd(&counters[16 + threadIdx.x], g_phAcc[threadIdx.x]);
#endif
  flush_stats(st, counters, STATS);
}

template <bool STATS, int LT>
__global__ __launch_bounds__(1024) void k_render_persist(DevScene S, RenderArgs A, float4* __restrict__ accum,
                                                         unsigned long long* __restrict__ counters) {
  persist_body<STATS, LT>(S, A, accum, counters);
}
// The same for the trees that leave the LDS to the stacks (LT_COMPACT3: the 424-word pool): workgroups of FOUR waves, one per
// SIMD, five of them per CU — 20 waves at 96 VGPRs (the compiler spills 28 registers, outside the descent: 1.4 % at equal
// occupancy on the 1 M-triangle scene, against what the four extra waves hide of its memory latency).
template <bool STATS, int LT>
__global__ __launch_bounds__(256, 5) void k_render_persist5(DevScene S, RenderArgs A, float4* __restrict__ accum,
                                                            unsigned long long* __restrict__ counters) {
  persist_body<STATS, LT>(S, A, accum, counters);
}

// Renderer.cpp:262-265
__global__ void k_resolve(uint32_t n_pixels, float spp, const float4* __restrict__ accum,
                          const float* __restrict__ bg, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  const float4 a = accum[i];
  const float miss = (float)((int)spp - (int)a.w);
  out[3 * i + 0] = a.x / spp + bg[3 * i + 0] * miss / spp;
  out[3 * i + 1] = a.y / spp + bg[3 * i + 1] * miss / spp;
  out[3 * i + 2] = a.z / spp + bg[3 * i + 2] * miss / spp;
}

// ---------------------------------------------------------------- ray streams (wavefront stage T)
// RayTracer::rayTrace (RayTracer.h:27-53) over a QUEUE of rays in HBM instead of the rays of
// one wave's pixels: persistent waves draw rays from a global counter — whenever enough
// lanes have finished, they are refilled in one step (ballot + mbcnt compaction, one atomic
// per refill) — so the 64 lanes of a wave stay busy however unequal the rays are, and the
// kernel carries no shading state: 64 VGPRs, 8 waves per SIMD.
//   rayO[i] = origin xyz, w = kind bits (bit 0: any-hit);  rayD[i] = direction xyz
//   res[i]  = closest: {t bits, triangle id} or {~0, ~0};  any-hit: {hit ? 1 : 0, 0}
constexpr int STREAM_CHUNK = 2048;
template <int LT>
__global__ __launch_bounds__(1024, 8) void k_trace_stream(DevScene S, const float4* __restrict__ rayO,
                                                          const float4* __restrict__ rayD, uint32_t n,
                                                          uint2* __restrict__ res, uint32_t* __restrict__ counter,
                                                          uint32_t stackLevels) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  if (LT != LT_NONE) {
    uint4* dst = reinterpret_cast<uint4*>(g_lds);
    for (uint32_t i = threadIdx.x; i < 2u * S.topK; i += blockDim.x) dst[i] = S.nodes[i];
    __syncthreads();
  }
  uint32_t* stack = g_lds + 8u * S.topK + wv * stackLevels * BLOCK + lane;
  Trav<TRAV_MIXED, LT> T;
  T.idle(stack);
  T.sharedKey = nullptr, T.pj = 0;
  LaneStats st;
  uint32_t mine = ~0u;
  // A wave reserves STREAM_CHUNK rays at a time (one atomic on the shared counter per chunk:
  // every wave of the chip hitting ONE address per refill was the bottleneck, 3 Grays/s) and
  // STAGES them 64 at a time in registers: lane L holds ray (block + L), loaded one refill
  // before it is needed, so the HBM/L2 latency of a ray record hides behind a traversal
  // round.  Lanes that finished take the staged rays in order, through a lane shuffle.
  uint32_t cur = 0, end = 0;      // unstaged rest of the chunk
  uint32_t sBase = 0, sHead = 64;  // staged block: ray index of entry 0, next entry to hand out
  uint32_t sCount = 0;             // entries of the staged block that exist
  float4 so = make_float4(0.f, 0.f, 0.f, 0.f), sd = so;
  bool more = true;  // rays left to stage
  for (;;) {
    const uint64_t idle = wave_ballot(!T.live());
    const int nIdle = __popcll(idle);
    if ((nIdle >= (int)S.refillT || nIdle == 64) && (sHead < sCount || more)) {
      if (sHead < sCount) {
        const uint32_t r = sHead + lanes_below(idle);
        const bool take = !T.live() && r < sCount;
        const int src = take ? (int)r : (int)lane;
        const float ox = __shfl(so.x, src, 64), oy = __shfl(so.y, src, 64), oz = __shfl(so.z, src, 64), ow = __shfl(so.w, src, 64);
        const float dx = __shfl(sd.x, src, 64), dy = __shfl(sd.y, src, 64), dz = __shfl(sd.z, src, 64);
        if (take) {
          T.start(mk(ox, oy, oz), mk(dx, dy, dz), S.invBoxScale);
          T.anyHit = (__float_as_uint(ow) & 1u) != 0u;
          mine = sBase + r;
          // (a ray with a NaN component is dead on arrival: Trav::start — it hits nothing)
          if (!T.live()) res[mine] = T.anyHit ? make_uint2(0u, 0u) : make_uint2(~0u, ~0u), mine = ~0u;
        }
        sHead = sHead + (uint32_t)nIdle < sCount ? sHead + (uint32_t)nIdle : sCount;
      }
      if (sHead >= sCount && more) {  // stage the next 64 (consumed at a later refill)
        if (cur == end) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(counter, (uint32_t)STREAM_CHUNK);
          cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          end = cur + (uint32_t)STREAM_CHUNK < n ? cur + (uint32_t)STREAM_CHUNK : n;
          if (cur >= n) cur = end = n, more = false;
        }
        if (more) {
          sBase = cur, sHead = 0, sCount = end - cur < 64u ? end - cur : 64u;
          if (lane < sCount) so = rayO[cur + lane], sd = rayD[cur + lane];
          cur += sCount;
        }
      }
    }
    if (wave_ballot(T.live()) == 0) {
      if (!more) break;
      continue;
    }
    const bool was = T.live();
    T.template round<false>(S, st);
    if (was && !T.live()) {
      res[mine] = T.anyHit ? make_uint2(T.found ? 1u : 0u, 0u)
                           : (T.found ? make_uint2(__float_as_uint(T.hit.t), T.hit.id) : make_uint2(~0u, ~0u));
      mine = ~0u;
    }
  }
}

// ---------------------------------------------------------------- wavefront integrator
#include "wavefront_kernels.h"

// ---------------------------------------------------------------- frame assembly (multi-GPU)
// A rank's owned 8x8-pixel granules, packed [granule][64 pixels] (row-major inside the
// granule; slots outside the image are never read back), and the inverse on the rank
// that assembles the frame.  gran[g] = x8 | y8 << 16 (granule coordinates).
__global__ void k_pack_owned(const float4* __restrict__ accum, float4* __restrict__ packed, const uint32_t* __restrict__ gran,
                             uint32_t n, uint32_t width, uint32_t height) {
  const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (g >= n) return;
  const uint32_t x = (gran[g] & 0xffffu) * 8u + (l & 7u), y = (gran[g] >> 16) * 8u + (l >> 3);
  packed[(size_t)g * 64u + l] = (x < width && y < height) ? accum[(size_t)y * width + x] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void k_unpack_owned(const float4* __restrict__ packed, float4* __restrict__ accum, const uint32_t* __restrict__ gran,
                               uint32_t n, uint32_t width, uint32_t height) {
  const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (g >= n) return;
  const uint32_t x = (gran[g] & 0xffffu) * 8u + (l & 7u), y = (gran[g] >> 16) * 8u + (l >> 3);
  if (x < width && y < height) accum[(size_t)y * width + x] = packed[(size_t)g * 64u + l];
}

// ---------------------------------------------------------------- test hooks
template <bool BRUTE, bool ANY, int LT = LT_NONE>
__global__ __launch_bounds__(BLOCK) void k_trace(DevScene S, const rt_ray* __restrict__ rays, uint32_t n,
                                                 rt_hit* __restrict__ hits, unsigned long long* counters) {
  __shared__ uint32_t lds[(STACK + 1) * BLOCK];  // (+1: the sentinel row)
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  LaneStats st;
  if (i < n) {
    HitRec h;
    const f3 o = ld(rays[i].origin), d = ld(rays[i].direction);
    // origins beyond the range the box padding was derived for (bvh_build.cpp: camera,
    // lights, geometry) are outside the exactness argument of the slab test: those rays
    // take the exhaustive loop (a NaN origin compares false and stays on the BVH path,
    // which ends it at once)
    const bool far = !BRUTE && fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) > S.originBound;
    bool found = cast<BRUTE, ANY, true, LT>(S, !far, o, d, lds + threadIdx.x, h, st);
    if (far) found = brute<ANY, true>(S, o, d, h, st);
    rt_hit r;
    r.hit = found, r.mesh = 0, r.tri = 0, r.vtx[0] = r.vtx[1] = r.vtx[2] = 0, r.u = r.v = r.d = 0.f;
    if (found && !ANY) {
      const uint4 tv = S.triShade[h.id];
      r.mesh = h.mesh;
      r.tri = h.id - S.meshTriBegin[h.mesh];
      const uint32_t vb = S.meshVtxBegin[h.mesh];
      r.vtx[0] = tv.x - vb, r.vtx[1] = tv.y - vb, r.vtx[2] = tv.z - vb;
      r.u = h.u, r.v = h.v, r.d = h.t;
    }
    hits[i] = r;
    if (ANY) st.shadow = 1;
    else st.closest = 1;
  }
  flush_stats(st, counters, true);
}

__global__ __launch_bounds__(BLOCK) void k_knn(DevScene S, const float* __restrict__ q, uint32_t n, uint32_t k,
                                               uint32_t* __restrict__ idx, float* __restrict__ dst,
                                               uint32_t* __restrict__ visited) {
  __shared__ uint32_t lds[(STACK + 2 * KMAX) * BLOCK];
  const Lds L = carve_lds<true>(lds);
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const Heap H{L.heap};
  const uint32_t vis = knn_query<false>(S, ld(q + 3 * (size_t)i), (int)k, H, L.stack);
  for (uint32_t j = 0; j < k; j++) {
    idx[(size_t)i * k + j] = H.I((int)j);
    dst[(size_t)i * k + j] = H.D((int)j);
  }
  if (visited) visited[i] = vis;
}

// PhotonMap::PhotonMap + calculatePhotonPath (PhotonMap.h:14-50,92-155), one lane
// per emitted photon, stream key (seed, RT_STREAM_PHOTON, light*perLight + j).
// Each emitted photon stores at most ONE particle, so slot j of the output is
// either that particle or flagged empty: order is deterministic (stable compaction on the
// device: kd_build.hip k_photon_compact, rt_build_photon_map; rt_emit_photons compacts on the host).
template <bool BRUTE>
__global__ __launch_bounds__(BLOCK) void k_emit(DevScene S, uint32_t perLight, uint32_t seed,
                                                float4* __restrict__ outPos, float4* __restrict__ outDir,
                                                unsigned long long* counters) {
  __shared__ uint32_t lds[(STACK + 1) * BLOCK];
  uint32_t* stack = lds + threadIdx.x;
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  LaneStats st;
  if (j < perLight * S.n_lights) {
    const uint32_t li = j / perLight;
    const rt_light Lt = S.lights[li];
    Rng g{rt_stream_seed(seed, RT_STREAM_PHOTON, j, 0)};
    const float lightPdf = 1.f / (float)S.n_lights;
    const f3 lsNormal = ld(Lt.normal);
    f3 o = light_sample(g, Lt);
    f3 d = hemisphere_sample(g, lsNormal);
    const float pdf0 = dot3(unit3(d), unit3(lsNormal));
    float weight = light_radiance(Lt, o) / (pdf0 * lightPdf);
    f3 ppos = mk(0.f, 0.f, 0.f), pdir = ppos;
    bool stored = false, exitNext = false;
    for (int depth = 0;; depth++) {
      if (exitNext) {
        stored = true;
        break;
      }
      if (depth >= 20) break;
      HitRec h;
      st.closest++;
      if (!cast<BRUTE, false, false>(S, true, o, d, stack, h, st)) {
        stored = depth != 0;
        break;
      }
      const float w = 1.f - h.u - h.v;
      const uint4 tv = S.triShade[h.id];
      const f3 nrm = unit3(interp3(S.vnrm, tv, w, h.u, h.v));
      const f3 pt = interp3(S.vpos, tv, w, h.u, h.v);
      ppos = pt, pdir = -d;
      const f3 rnd = hemisphere_sample(g, nrm);
      const f3 perfect = d - 2.f * (dot3(d, nrm)) * nrm;
      const float bsdf = len3(bsdf_eval(S.mats[h.mesh], nrm, d, rnd));
      const float pdf = (dot3(unit3(rnd), unit3(perfect)) + 1.f) / 2.f;
      weight *= bsdf / pdf;
      const float cont = fminf(weight, 1.f);
      if (g.uniformF(0.f, 1.f) > cont) exitNext = true;
      else weight /= cont;
      o = pt, d = rnd;
    }
    outPos[j] = make_float4(ppos.x, ppos.y, ppos.z, stored ? 1.f : 0.f);
    outDir[j] = make_float4(pdir.x, pdir.y, pdir.z, weight);
  }
  flush_stats(st, counters, false);
}

// Unit evaluations of the device building blocks (parity hooks, rt_test_unit).
__global__ void k_unit(uint32_t which, const void* __restrict__ in, void* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (which) {
    case RT_UNIT_ASIN:
      ((double*)out)[i] = rt_asin(((const double*)in)[i]);
      break;
    case RT_UNIT_SINF:
      ((float*)out)[i] = rt_sinf(((const float*)in)[i]);
      break;
    case RT_UNIT_COSF:
      ((float*)out)[i] = rt_cosf(((const float*)in)[i]);
      break;
    case RT_UNIT_STREAM_SEED: {
      const uint32_t* a = (const uint32_t*)in + 4 * (size_t)i;
      ((uint32_t*)out)[i] = rt_stream_seed(a[0], a[1], a[2], a[3]);
      break;
    }
    case RT_UNIT_TRIANGLE: {  // in: p0 p1 p2 o d (15 floats); out: hit u v t (4 floats)
      const float* a = (const float*)in + 15 * (size_t)i;
      float* o = (float*)out + 4 * (size_t)i;
      const f3 p0 = ld(a), p1 = ld(a + 3), p2 = ld(a + 6);
      float u = o[1], v = o[2], t = o[3];  // outputs may stay unwritten (Ray.cpp:14)
      const bool hit = tri_test_ref_order(ld(a + 9), ld(a + 12), p0, p1 - p0, p2 - p0, u, v, t);
      o[0] = hit ? 1.f : 0.f, o[1] = u, o[2] = v, o[3] = t;
      break;
    }
    case RT_UNIT_BSDF: {  // in: kd alpha albedo3 f03 n3 wi3 wo3 (17 floats); out: 3 floats
      const float* a = (const float*)in + 17 * (size_t)i;
      rt_material m;
      m.kd = a[0], m.alpha = a[1];
      for (int c = 0; c < 3; c++) m.albedo[c] = a[2 + c], m.f0[c] = a[5 + c];
      const f3 r = bsdf_eval(m, ld(a + 8), ld(a + 11), ld(a + 14));
      float* o = (float*)out + 3 * (size_t)i;
      o[0] = r.x, o[1] = r.y, o[2] = r.z;
      break;
    }
    case RT_UNIT_RAY_AT: {  // in: camera(12) u v (14 floats); out: o3 d3
      const float* a = (const float*)in + 14 * (size_t)i;
      rt_camera c;
      for (int k = 0; k < 3; k++)
        c.position[k] = a[k], c.lower_left[k] = a[3 + k], c.horizontal[k] = a[6 + k], c.vertical[k] = a[9 + k];
      f3 o, d;
      camera_ray(c, a[12], a[13], o, d);
      float* q = (float*)out + 6 * (size_t)i;
      q[0] = o.x, q[1] = o.y, q[2] = o.z, q[3] = d.x, q[4] = d.y, q[5] = d.z;
      break;
    }
    case RT_UNIT_LIGHT_EVAL: {  // in: rt_light(21 floats) p3; out: 3
      const float* a = (const float*)in + 24 * (size_t)i;
      rt_light l;
      memcpy(&l, a, sizeof(rt_light));
      const f3 r = light_eval(l, ld(a + 21));
      float* o = (float*)out + 3 * (size_t)i;
      o[0] = r.x, o[1] = r.y, o[2] = r.z;
      break;
    }
    case RT_UNIT_SAMPLERS: {
      // in (28 words): state, idx, n, pad, normal3, pad, rt_light(21 floats).
      // out (12 words): jitter x y, hemisphere dir3, light sample3, end state, pad3.
      // Draw order: jitter, hemisphere, light — each from the running state.
      const uint32_t* a = (const uint32_t*)in + 28 * (size_t)i;
      const float* af = (const float*)a;
      Rng g{a[0]};
      float* o = (float*)out + 12 * (size_t)i;
      jitter_sample(g, (int)a[1], (int)a[2], o[0], o[1]);
      const f3 h = hemisphere_sample(g, ld(af + 4));
      rt_light l;
      memcpy(&l, af + 7, sizeof(rt_light));
      const f3 p = light_sample(g, l);
      o[2] = h.x, o[3] = h.y, o[4] = h.z, o[5] = p.x, o[6] = p.y, o[7] = p.z;
      ((uint32_t*)o)[8] = g.s;
      o[9] = o[10] = o[11] = 0.f;
      break;
    }
    case RT_UNIT_LIGHT_SAMPLE: {  // in (22 words): state, rt_light(21 floats); out: sample3, end state
      const uint32_t* a = (const uint32_t*)in + 22 * (size_t)i;
      Rng g{a[0]};
      rt_light l;
      memcpy(&l, a + 1, sizeof(rt_light));
      const f3 p = light_sample(g, l);
      float* o = (float*)out + 4 * (size_t)i;
      o[0] = p.x, o[1] = p.y, o[2] = p.z;
      ((uint32_t*)o)[3] = g.s;
      break;
    }
    case RT_UNIT_POW: {  // in: double x; out: double rt_pow2(x), rt_pow5(x)
      const double x = ((const double*)in)[i];
      ((double*)out)[2 * (size_t)i] = rt_pow2(x), ((double*)out)[2 * (size_t)i + 1] = rt_pow5(x);
      break;
    }
    case RT_UNIT_RECIP: {  // in: float x; out: rtd::recip_fast(x), 1.0f / x, rtd::sqrt_fast(x), sqrtf(x)
      const float x = ((const float*)in)[i];
      float* o = (float*)out + 4 * (size_t)i;
      o[0] = rtd::recip_fast(x), o[1] = 1.0f / x, o[2] = rtd::sqrt_fast(x), o[3] = __builtin_sqrtf(x);
      break;
    }
    default:
      break;
  }
}

// ---------------------------------------------------------------- launchers
// Dynamic LDS beyond 64 KiB must be allowed per kernel AND per device (one process may drive
// several devices: rt_group): `done` is the caller's per-kernel bit mask of devices already set.
template <class K>
static bool allow_big_lds(K kernel, unsigned long long& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  if (done & (1ull << dev)) return true;
#ifndef RT_PHASE_TIMING
  // Trav::round takes a node's byte offset as its LDS address: the tree copy must start at LDS
  // address 0, i.e. the kernel must not have static LDS in front of the dynamic segment
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) != hipSuccess || fa.sharedSizeBytes != 0) return false;
#endif
  // (the diagnostic build stays 256 B short of the CU's 160 KiB for its static LDS: the attribute
  // is refused when static + dynamic exceed the CU)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             4 * rtbvh::kLdsWordsPerCU);
  done |= 1ull << dev;
  return true;
}

// LDS plan of the persistent pooled kernel: W waves per workgroup (one workgroup per CU),
// topK tree-top nodes in front of W private regions of waveWords each.  160 KiB per CU.
struct PersistPlan {
  uint32_t waves, topK, waveWords, ldsBytes;
  int compact = 0;  // 1: LT_COMPACT pool layout, 2: LT_COMPACT2, 3: LT_COMPACT3 (workgroups of 4 waves, 5 per CU)
};
static PersistPlan plan_persist(const DevScene& S, const RenderArgs& A) {
  const uint32_t total = rtbvh::kLdsWordsPerCU;  // words
  const uint32_t waveWords = A.stackLevels * BLOCK + VP_WORDS;
  static const int wEnv = getenv("RT_PERSIST_WAVES") ? atoi(getenv("RT_PERSIST_WAVES")) : 0;
  static const int kEnv = getenv("RT_TOPK") ? atoi(getenv("RT_TOPK")) : -1;
  const uint32_t cap = S.n_nodes < rtbvh::kTopNodes ? S.n_nodes : rtbvh::kTopNodes;
  // Measured (MI355X, Grays/s): occupancy comes first — 12 instead of 16 waves costs 15-17 %
  // on every scene; the whole tree in LDS (LT_ALL, no branch) gives C2 +5.6 %.  So: as many
  // waves as fit (at most 16), then as much of the tree as fits beside them.
  // RT_PERSIST_WAVES / RT_TOPK override.
  uint32_t w = wEnv > 0 ? (uint32_t)wEnv : 16u;
  while (w > 1u && w * waveWords > total) --w;
  if (w * waveWords > total) return PersistPlan{0, 0, waveWords, 0};
  // big trees whose stacks leave fewer than 16 waves: the compact pool, if it buys a wave
  static const int cpEnv = getenv("RT_COMPACT") ? atoi(getenv("RT_COMPACT")) : -1;
  // ... and if the smallest pool makes room for TWENTY waves (five workgroups of four per CU, 96 VGPRs): the 1 M-triangle
  // scene's 25 stack rows + 424 words = 8,096 B per wave
  if ((cpEnv < 0 || cpEnv == 3) && S.n_nodes > kPrioMaxNodes && !S.q8 && wEnv <= 0 && kEnv < 0) {
    // (the LDS is handed out in units of 1,280 bytes: five workgroups of four waves must fit with their sizes rounded up)
    const uint32_t ww3 = A.stackLevels * BLOCK + (uint32_t)VpLayout<3>::WORDS;
    if (5u * ((16u * ww3 + rtbvh::kLdsGrainBytes - 1u) / rtbvh::kLdsGrainBytes * rtbvh::kLdsGrainBytes) <= 4u * total) {
      PersistPlan cp{4u, 0u, ww3, 4u * 4u * ww3};
      cp.compact = 3;
      return cp;
    }
  }
  if (cpEnv != 0 && (S.n_nodes > kPrioMaxNodes || S.q8) && w < 16u && kEnv < 0) {
    // (the smallest step that buys the most waves; RT_COMPACT = 1 / 2 forces a level)
    uint32_t bestW = w;
    int level = 0;
    for (int lv = 1; lv <= 2; ++lv) {
      const uint32_t ww = waveWords - (uint32_t)(lv * VP_COMPACT_SAVES);
      uint32_t w2 = wEnv > 0 ? (uint32_t)wEnv : 16u;
      while (w2 > 1u && w2 * ww > total) --w2;
      if (cpEnv > 0 ? lv == cpEnv : w2 > bestW) bestW = w2, level = lv;
    }
    if (level) {
      const uint32_t ww = waveWords - (uint32_t)(level * VP_COMPACT_SAVES);
      PersistPlan cp{bestW, 0, ww, 4u * bestW * ww};
      cp.compact = level;
      return cp;
    }
  }
  uint32_t k = S.q8 ? 0u : (total - w * waveWords) / 8u;  // (Q8 records are not copied into LDS)
  k = k < cap ? k : cap;
  // (a PARTIAL top — a prefix of the area-ordered node array — costs the step a second load path;
  // with the step as lean as it is now it pays from a few hundred nodes up: the 11.7 k-triangle
  // mesh keeps 688 of its 6,003 nodes in LDS beside 16 waves, +1.8 %; 256 nodes +0.7 %)
  if (kEnv >= 0) k = (uint32_t)kEnv < k ? (uint32_t)kEnv : k;
  else if (k < S.n_nodes && k < 256u) k = 0;
  return PersistPlan{w, k, waveWords, 4u * (8u * k + w * waveWords)};
}

template <bool BRUTE, bool PHOTON, bool POOLED>
static hipError_t launch_render2(bool stats, const DevScene& S, const RenderArgs& A, float4* accum,
                                 unsigned long long* counters, hipStream_t stream) {
  const uint32_t blocks = A.n_tiles;
  if (blocks == 0) return hipSuccess;
  static const bool noPersist = getenv("RT_NO_PERSIST") != nullptr;
  if (POOLED && !noPersist && A.tileCounter && A.numCUs) {
    const PersistPlan P = plan_persist(S, A);
    if (P.waves) {
      DevScene S2 = S;
      RenderArgs A2 = A;
      S2.topK = P.topK, A2.waveWords = P.waveWords;
      hipError_t e = hipMemsetAsync(A.tileCounter, 0, sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
      const uint32_t perCU = P.waves;                                    // waves one workgroup brings
      const uint32_t slots = P.compact == 3 ? 5u * A.numCUs : A.numCUs;  // workgroups the device holds at once
      const uint32_t wgs = (blocks + perCU - 1) / perCU < slots ? (blocks + perCU - 1) / perCU : slots;
#define RT_LAUNCH_PERSIST(ST, LTV)                                                                          \
  do {                                                                                                      \
    static unsigned long long done = 0;                                                                     \
    if (!allow_big_lds(&k_render_persist<ST, LTV>, done)) return hipErrorInvalidConfiguration;              \
    hipLaunchKernelGGL((k_render_persist<ST, LTV>), dim3(wgs), dim3(64u * P.waves), P.ldsBytes, stream, S2, \
                       A2, accum, counters);                                                                \
  } while (0)
      // Two instances per layout: the timed one, with the three-instruction 1 / det and 1 / length (the host has bounded
      // their operands: DevScene::slowRecip == 0), and the COUNTED one, which divides — and which also serves the rare
      // scene beyond those bounds (same frame: the short forms give the division's bits).
#define RT_LAUNCH_EITHER(LTV)                                    \
  do {                                                           \
    if (stats || S.slowRecip) RT_LAUNCH_PERSIST(true, LTV);      \
    else RT_LAUNCH_PERSIST(false, (LTV) | LT_FASTDET);           \
  } while (0)
      const int lt = P.topK == 0 ? LT_NONE : P.topK >= S.n_nodes ? LT_ALL : LT_TOP;
      if (P.compact == 3) {
        static unsigned long long done5s = 0, done5t = 0;
        if (stats || S.slowRecip) {
          if (!allow_big_lds(&k_render_persist5<true, LT_NONE | LT_NOPRIO | LT_COMPACT3>, done5s)) return hipErrorInvalidConfiguration;
          hipLaunchKernelGGL((k_render_persist5<true, LT_NONE | LT_NOPRIO | LT_COMPACT3>), dim3(wgs), dim3(256), P.ldsBytes, stream, S2, A2, accum, counters);
        } else {
          if (!allow_big_lds(&k_render_persist5<false, LT_NONE | LT_NOPRIO | LT_COMPACT3 | LT_FASTDET>, done5t)) return hipErrorInvalidConfiguration;
          hipLaunchKernelGGL((k_render_persist5<false, LT_NONE | LT_NOPRIO | LT_COMPACT3 | LT_FASTDET>), dim3(wgs), dim3(256), P.ldsBytes, stream, S2, A2,
                             accum, counters);
        }
      } else if (S.q8) {  // the one-request records (the context chose them: rt_api.cpp create_ctx)
        if (P.compact == 2) RT_LAUNCH_EITHER(LT_Q8 | LT_NOPRIO | LT_COMPACT2);
        else if (P.compact) RT_LAUNCH_EITHER(LT_Q8 | LT_NOPRIO | LT_COMPACT);
        else RT_LAUNCH_EITHER(LT_Q8 | LT_NOPRIO);
      } else if (P.compact == 2) RT_LAUNCH_EITHER(LT_NONE | LT_NOPRIO | LT_COMPACT2);
      else if (P.compact) RT_LAUNCH_EITHER(LT_NONE | LT_NOPRIO | LT_COMPACT);
      else if (lt == LT_ALL) RT_LAUNCH_EITHER(LT_ALL);
      else if (lt == LT_TOP) RT_LAUNCH_EITHER(LT_TOP);
      else RT_LAUNCH_EITHER(LT_NONE | LT_NOPRIO);  // (a small tree beside stacks that leave no room for 256 of its nodes: rare)
#undef RT_LAUNCH_EITHER
#undef RT_LAUNCH_PERSIST
      return hipGetLastError();
    }
  }
  const uint32_t rows = A.stackLevels + (PHOTON ? 2 * A.k : 0);
  const size_t ldsBytes = 4u * ((rows < 4u ? 4u : rows) * BLOCK + (POOLED ? VP_WORDS : 0));
  RenderArgs A1 = A;
  A1.tilesPerBlock = 1u;  // (more wave tiles per workgroup unbalance the grid: C3 13.7 / 14.8 / 17.8 ms at 4 / 8 / 16 in round 3)
  const uint32_t nBlocks = blocks;
  // Occupancy target (waves per SIMD) of the one-wave-per-workgroup kernels.
  // (the photon kernel at 5 waves per SIMD: its walk waits on L1-hit loads half of its life; 96 VGPRs cost 12 spilled registers
  // outside the walk, and the 16-bit stack makes the LDS room — measured: C3 7.3 -> see DESIGN.md 4.6)
  constexpr int MINW = PHOTON ? 5 : 4;
  if (stats) hipLaunchKernelGGL((k_render<BRUTE, PHOTON, POOLED, true, 1>), dim3(nBlocks), dim3(BLOCK), ldsBytes, stream, S, A1, accum, counters);
  else {
    hipLaunchKernelGGL((k_render<BRUTE, PHOTON, POOLED, false, MINW>), dim3(nBlocks), dim3(BLOCK), ldsBytes, stream, S, A1, accum, counters);
    hipLaunchKernelGGL(k_fold_stripes, dim3(1), dim3(1024), 0, stream, counters);
  }
  return hipGetLastError();
}

hipError_t launch_render(bool brute_force, bool photon, bool stats, const DevScene& S, const RenderArgs& A,
                         float4* accum, unsigned long long* counters, hipStream_t stream) {
  if (brute_force) return photon ? launch_render2<true, true, false>(stats, S, A, accum, counters, stream)
                                 : launch_render2<true, false, false>(stats, S, A, accum, counters, stream);
  if (photon) return launch_render2<false, true, false>(stats, S, A, accum, counters, stream);
  // direct lighting through the BVH: the vertex pool handles up to POOL_L lights
  if ((A.flags & 1u) && S.n_lights <= (uint32_t)POOL_L) return launch_render2<false, false, true>(stats, S, A, accum, counters, stream);
  return launch_render2<false, false, false>(stats, S, A, accum, counters, stream);
}

hipError_t launch_resolve(uint32_t n_pixels, uint32_t spp, const float4* accum, const float* bg, float* out,
                          hipStream_t stream) {
  if (n_pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(k_resolve, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, n_pixels, (float)spp, accum,
                     bg, out);
  return hipGetLastError();
}

hipError_t launch_trace_stream(const DevScene& S, const float4* rayO, const float4* rayD, uint32_t n, uint2* res,
                               uint32_t* counter, uint32_t stackLevels, uint32_t numCUs, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  // LDS: [tree, if all of it fits beside the stacks][16 waves x stack]; two workgroups per CU
  const uint32_t total = 160u * 1024u / 4u / 2u - 32u;  // words per workgroup
  uint32_t wpw = 16u;                                   // waves per workgroup: as many as the stacks allow
  while (wpw > 1u && wpw * stackLevels * BLOCK > total) --wpw;
  const uint32_t stackWords = wpw * stackLevels * BLOCK;
  static const int kEnv = getenv("RT_STREAM_TOPK") ? atoi(getenv("RT_STREAM_TOPK")) : -1;
  DevScene S2 = S;
  S2.topK = 0;
  if (stackWords > total) return hipErrorInvalidValue;
  if (stackWords + 8u * S.n_nodes <= total) S2.topK = S.n_nodes;
  if (kEnv >= 0) S2.topK = (uint32_t)kEnv < S2.topK ? (uint32_t)kEnv : S2.topK;
  const uint32_t ldsBytes = 4u * (8u * S2.topK + stackWords);
  const uint32_t waves = (n + 63u) / 64u;
  uint32_t wgs = (waves + wpw - 1u) / wpw;
  static const uint32_t perCU = getenv("RT_STREAM_WGS") ? (uint32_t)atoi(getenv("RT_STREAM_WGS")) : 2u;
  wgs = wgs < perCU * numCUs ? wgs : perCU * numCUs;
  if (S2.topK >= S.n_nodes && S2.topK) {
    static unsigned long long done = 0;
    if (!allow_big_lds(&k_trace_stream<LT_ALL>, done)) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((k_trace_stream<LT_ALL>), dim3(wgs), dim3(64u * wpw), ldsBytes, stream, S2, rayO, rayD, n, res, counter, stackLevels);
  } else {
    static unsigned long long done = 0;
    if (!allow_big_lds(&k_trace_stream<LT_NONE>, done)) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((k_trace_stream<LT_NONE>), dim3(wgs), dim3(64u * wpw), ldsBytes, stream, S2, rayO, rayD, n, res, counter, stackLevels);
  }
  return hipGetLastError();
}

hipError_t launch_pack(bool unpack, const float4* src, float4* dst, const uint32_t* gran, uint32_t n, uint32_t width,
                       uint32_t height, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + 3) / 4), block(256);
  if (unpack) hipLaunchKernelGGL(k_unpack_owned, grid, block, 0, stream, src, dst, gran, n, width, height);
  else hipLaunchKernelGGL(k_pack_owned, grid, block, 0, stream, src, dst, gran, n, width, height);
  return hipGetLastError();
}


hipError_t launch_trace(bool brute_force, bool any, const DevScene& S, const rt_ray* rays, uint32_t n,
                        rt_hit* hits, unsigned long long* counters, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + BLOCK - 1) / BLOCK), block(BLOCK);
  if (brute_force) {
    if (any) hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, stream, S, rays, n, hits, counters);
    else hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, stream, S, rays, n, hits, counters);
  } else if (S.q8) {  // the context traverses the one-request records
    if (any) hipLaunchKernelGGL((k_trace<false, true, LT_Q8>), grid, block, 0, stream, S, rays, n, hits, counters);
    else hipLaunchKernelGGL((k_trace<false, false, LT_Q8>), grid, block, 0, stream, S, rays, n, hits, counters);
  } else {
    if (any) hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, stream, S, rays, n, hits, counters);
    else hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, stream, S, rays, n, hits, counters);
  }
  return hipGetLastError();
}

hipError_t launch_knn(const DevScene& S, const float* q, uint32_t n, uint32_t k, uint32_t* idx, float* dist,
                      uint32_t* visited, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_knn, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, S, q, n, k, idx, dist, visited);
  return hipGetLastError();
}

hipError_t launch_emit(const DevScene& S, uint32_t perLight, uint32_t seed, float4* outPos, float4* outDir,
                       unsigned long long* counters, hipStream_t stream) {
  const uint32_t n = perLight * S.n_lights;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_emit<false>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, S, perLight, seed,
                     outPos, outDir, counters);
  return hipGetLastError();
}

// recip_fast / sqrt_fast against the division and sqrtf on THIS device: every mantissa (2^23) in two neighbouring
// binades (both signs for the reciprocal) — v_rcp_f32 and v_rsq_f32 work on the mantissa and the exponent's parity, and
// scaling by powers of four is exact on both sides of the comparison away from the denormals, so this covers what the
// exhaustive check of tools/microbench/recip_exact.hip found to be the whole in-range behaviour.  *bad != 0: a mismatch.
__global__ void k_selfcheck_recip(uint32_t* bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // 2^25: sign | binade | mantissa
  const uint32_t bits = ((i >> 24) << 31) | ((127u + ((i >> 23) & 1u)) << 23) | (i & 0x7fffffu);
  const float x = __uint_as_float(bits);
  bool ok = __float_as_uint(rtd::recip_fast(x)) == __float_as_uint(1.0f / x);
  if (!(i >> 24)) ok = ok && __float_as_uint(rtd::sqrt_fast(x)) == __float_as_uint(__builtin_sqrtf(x));
  if (!ok) atomicOr(bad, 1u);
}
hipError_t launch_selfcheck_recip(uint32_t* dBad, hipStream_t stream) {
  hipLaunchKernelGGL(k_selfcheck_recip, dim3(1u << 17), dim3(256), 0, stream, dBad);
  return hipGetLastError();
}

hipError_t launch_unit(uint32_t which, const void* in, void* out, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unit, dim3((n + 255) / 256), dim3(256), 0, stream, which, in, out, n);
  return hipGetLastError();
}

}  // namespace rtk
