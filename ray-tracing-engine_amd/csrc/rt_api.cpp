// rt_api.cpp — the C ABI declared in include/rt_amd.h: context lifetime, HBM
// residency of the flattened scene, launch orchestration.  All arithmetic of the
// hot path lives in rt_kernels.hip; nothing here computes radiance on the host
// and nothing falls back to the CPU when a device is missing.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "rt_amd.h"
#include "rt_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Picks the device and refuses anything that is not a gfx950 part.
int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(RT_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RT_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));
  return RT_OK;
}

template <class T>
int upload(T** dptr, const void* src, size_t count) {
  *dptr = nullptr;
  if (count == 0) return RT_OK;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), count * sizeof(T)));
  HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
  return RT_OK;
}

struct TileKey {
  uint32_t w = 0, h = 0, rank = 0, world = 0, tile = 0, sshift = 0;
  bool operator==(const TileKey& o) const {
    return w == o.w && h == o.h && rank == o.rank && world == o.world && tile == o.tile && sshift == o.sshift;
  }
};

// A wave integrates (64 >> sshift) pixels x (1 << sshift) samples at a time.  Pixel
// footprint of one wave for sshift = 0..6: 8x8, 8x4, 4x4, 4x2, 2x2, 2x1, 1x1.
void wave_tile_shape(uint32_t sshift, uint32_t& tw, uint32_t& th) {
  static const uint32_t W[7] = {8, 8, 4, 4, 2, 2, 1}, H[7] = {8, 4, 4, 2, 2, 1, 1};
  tw = W[sshift], th = H[sshift];
}

constexpr int kEventPairs = 256;

}  // namespace

struct rt_ctx {
  int device = 0;
  rtk::DevScene S{};
  rtbvh::Built bvh;  // host copy kept for rt_bvh_export
  std::vector<void*> allocs;
  float4* phPos = nullptr;
  float4* phDir = nullptr;
  uint4* phTopo = nullptr;  // explicit kd topology over phPos (rtk::launch_kd_topology)
  uint32_t* dTiles = nullptr;
  uint32_t nTiles = 0;
  TileKey tileKey;
  unsigned long long* dCounters = nullptr;
  uint32_t* dTileCounter = nullptr;  // work queue head of the persistent render kernel
  // owned-granule lists of the ranks of a tile-sharded frame (multi-GPU assembly), by key
  struct GranList {
    uint32_t* d = nullptr;
    uint32_t n = 0;
  };
  std::map<std::string, GranList> granules;
  // device-built BVH (rt_options.bvh_builder / RT_BVH_GPU): the float form of the nodes stays
  // on the device for rt_bvh_export
  float4* dNodesF = nullptr;
  // path state + ray queues of the wavefront integrator (allocated on first use)
  rtk::WfArgs wf{};
  size_t wfCap = 0;
  void* wfBlock = nullptr;
  uint32_t builder = RT_BVH_HOST;
  bool broken = false;  // the device tree is in an unknown state (rt_bvh_tune could not restore it): launches are refused
  uint32_t recipCheck = 0;  // 0: short reciprocal forms not wanted (operand bounds), 1: verified on this device, 2: self-check FAILED (dividing)
  uint32_t nodeFormat = RT_NODES_F16;  // what the pooled render kernel and rt_trace traverse
  float buildMs = 0.f;
  uint32_t numCUs = 0;
  hipEvent_t ev[kEventPairs][2];
  int evUsed = 0;
  bool evReady = false;
};

namespace {

template <class T>
int upload_owned(rt_ctx* c, const T** field, const void* src, size_t count) {
  T* d = nullptr;
  int rc = upload(&d, src, count);
  if (rc != RT_OK) return rc;
  if (d) c->allocs.push_back(d);
  *field = d;
  return RT_OK;
}

// How many samples of a pixel one wave integrates side by side.  A pixel's float
// sum must be formed in sample order, but only the ADDS are ordered: the samples
// themselves are independent streams, so the 64 lanes of a wave can hold
// (pixel, sample) pairs and hand their results to the pixel's owner lane, which adds
// them in order.  More samples per wave = more, shorter work items: the grid no longer
// quantises into ~3 rounds of 16k tile-sized items, and a rank that owns 1/8 of the
// pixels still fills the GPU.
uint32_t choose_sshift(const rt_ctx* c, const rt_params* p, uint32_t spp_count) {
  if (p->reserved[0]) {  // explicit lanes-per-pixel (tests, experiments)
    uint32_t s = 0;
    while ((1u << (s + 1)) <= p->reserved[0] && s < 6) ++s;
    return s;
  }
  const uint64_t world = p->world ? p->world : 1;
  const uint64_t pixels = (uint64_t)p->width * p->height / world;
  // measured on C2 (1024^2 x 128 spp), samples per wave 1/2/4/8/16/64:
  // 12.3 / 12.8 / 13.4 / 13.6 / 13.7 / 13.6 Grays/s -> aim for >= 32 rounds of a
  // 256-CU x 16-wave chip
  const uint64_t target = 32ull * 256 * 16;
  uint32_t s = 0;
  while (s < 6 && (pixels << s) / 64 < target && (2u << s) <= spp_count) ++s;
  // The pooled BVH kernel (persistent workgroups: no grid quantisation to balance) wants MORE samples of a pixel side by
  // side than that: their vertices lie close together, so the wave's rays share nodes and triangle lines.  Measured in
  // round 3 (profiles/r03_samples_per_wave.txt), frame ms at 1 / 4 / 8 / 16 / 32 / 64 samples of a pixel per wave:
  // C2 56.1 / 51.2 / 50.4 / 50.3 / 50.6 / 51.4, C4 - / 937.6 / 934.0 / 931.5 / - / 951.8 (the old rule gave it 2: 946),
  // and the scenes that sit on the vector L1's request roof, where coalescing is worth most: 1 M triangles 407 / 358 /
  // 348 / 339 / 334 / 328, 8 M triangles (32 spp) - / 59.7 / 57.6 / 55.8 / 55.1.  So: 16 on cache-resident scenes, as
  // many as the frame has (<= 64) beyond 65,536 nodes.  (The owner lane's in-order adds grow with the count: that is
  // what turns C2 and C4 around after 16.)  The image does not depend on it (test_frame_independent_of_samples_per_wave).
  const bool pooled = !p->use_photons && p->accel != RT_ACCEL_BRUTE && c->S.n_lights <= 3u && !(p->reserved[1] & 1u) &&
                      !(p->reserved[2] & 1u);
  if (pooled) {
    const uint32_t want = c->S.n_nodes > 65536u ? 6u : 4u;
    while (s < want && (2u << s) <= spp_count) ++s;
  }
  // Photon-map shading: the k-NN walks of a pixel's samples run almost in step (their queries lie within one pixel's
  // footprint), and a wave issues every branch any of its lanes is in — C3 at 1 / 2 / 4 / 8 / 16 samples of a pixel per
  // wave: 26.3 / 19.6 / 15.7 / 13.7 / 12.4 ms (16 was 17.0 ms until the waves' counts went to striped slots:
  // rt_kernels.hip flush_stats_striped); at 64 spp, 16 / 32 / 64: 46.6 / 43.9 / 42.8 ms (profiles/r03_c3_samples_per_wave.txt)
  if (p->use_photons && p->accel != RT_ACCEL_BRUTE)
    while (s < 6u && (2u << s) <= spp_count) ++s;
  return s;
}

int ensure_tiles(rt_ctx* c, const rt_params* p, uint32_t sshift) {
  TileKey k;
  k.w = p->width, k.h = p->height, k.rank = p->rank, k.world = p->world ? p->world : 1;
  k.tile = p->tile ? p->tile : 8;
  k.sshift = sshift;
  if (c->dTiles && k == c->tileKey) return RT_OK;
  std::vector<uint32_t> tiles;
  uint32_t tw, th;
  wave_tile_shape(sshift, tw, th);
  // enumerate 8x8 granules row-major and the wave tiles inside each granule, so that
  // consecutive waves touch neighbouring pixels
  const uint32_t gx = (k.w + 7) / 8, gy = (k.h + 7) / 8;
  for (uint32_t y8 = 0; y8 < gy; ++y8)
    for (uint32_t x8 = 0; x8 < gx; ++x8) {
      const uint32_t ox = x8 * 8 / k.tile, oy = y8 * 8 / k.tile;
      if (k.world > 1 && (ox + oy) % k.world != k.rank) continue;
      for (uint32_t y = y8 * 8; y < y8 * 8 + 8 && y < k.h; y += th)
        for (uint32_t x = x8 * 8; x < x8 * 8 + 8 && x < k.w; x += tw) tiles.push_back(x | (y << 16));
    }
  if (c->dTiles) {
    HIP_TRY(hipFree(c->dTiles));
    c->dTiles = nullptr;
  }
  int rc = upload(&c->dTiles, tiles.data(), tiles.size());
  if (rc != RT_OK) return rc;
  c->nTiles = static_cast<uint32_t>(tiles.size());
  c->tileKey = k;
  return RT_OK;
}

// The 8x8-pixel granules rank `rank` of `world` owns, row-major — the order ensure_tiles
// renders them in and the order of the packed exchange buffer.
void owned_granules(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t tile, std::vector<uint32_t>* out,
                    uint32_t* count) {
  if (tile == 0) tile = 8;
  if (world == 0) world = 1;
  uint32_t n = 0;
  const uint32_t gx = (w + 7) / 8, gy = (h + 7) / 8;
  for (uint32_t y8 = 0; y8 < gy; ++y8)
    for (uint32_t x8 = 0; x8 < gx; ++x8) {
      if (world > 1 && (x8 * 8 / tile + y8 * 8 / tile) % world != rank) continue;
      if (out) out->push_back(x8 | (y8 << 16));
      ++n;
    }
  if (count) *count = n;
}

int ensure_granules(rt_ctx* c, const rt_params* p, uint32_t rank, rt_ctx::GranList* out) {
  char key[96];
  snprintf(key, sizeof key, "%u.%u.%u.%u.%u", p->width, p->height, rank, p->world ? p->world : 1, p->tile ? p->tile : 8);
  auto it = c->granules.find(key);
  if (it == c->granules.end()) {
    std::vector<uint32_t> g;
    owned_granules(p->width, p->height, rank, p->world, p->tile, &g, nullptr);
    rt_ctx::GranList L;
    int rc = upload(&L.d, g.data(), g.size());
    if (rc != RT_OK) return rc;
    L.n = static_cast<uint32_t>(g.size());
    it = c->granules.emplace(key, L).first;
  }
  *out = it->second;
  return RT_OK;
}

int check_params(const rt_ctx* c, const rt_params* p) {
  if (!p) return fail(RT_ERR_INVALID, "params is null");
  if (c->broken) return fail(RT_ERR_STATE, "the context's device tree is in an unknown state (a failed rt_bvh_tune): destroy it");
  if (p->width == 0 || p->height == 0 || p->width > 65535u || p->height > 65535u)
    return fail(RT_ERR_INVALID, "image size %ux%u out of range", p->width, p->height);
  if (p->spp == 0) return fail(RT_ERR_INVALID, "spp must be >= 1");
  if (p->mode != RT_MODE_RAY && p->mode != RT_MODE_PATH) return fail(RT_ERR_INVALID, "mode must be 0 or 1");
  if (p->rng_mode != RT_RNG_PIXEL)
    return fail(RT_ERR_UNSUPPORTED,
                "rng_mode legacy is one global serial engine (reference LightSource.h:6) and cannot run in "
                "parallel; the GPU path implements RT_RNG_PIXEL only");
  if (p->max_depth < 1 || p->max_depth > 3) return fail(RT_ERR_UNSUPPORTED, "max_depth must be in 1..3");
  if (p->world > 1 && p->rank >= p->world) return fail(RT_ERR_INVALID, "rank %u >= world %u", p->rank, p->world);
  if (p->tile % 8 != 0) return fail(RT_ERR_INVALID, "tile must be a multiple of 8");
  if (p->spp_count && (uint64_t)p->spp_begin + p->spp_count > p->spp)
    return fail(RT_ERR_INVALID, "sample range [%u,+%u) exceeds spp %u", p->spp_begin, p->spp_count, p->spp);
  if (p->use_photons) {
    if (c->S.n_photons == 0) return fail(RT_ERR_STATE, "use_photons set but no photons were uploaded (rt_set_photons)");
    if (p->k < 1 || p->k > RTK_KMAX) return fail(RT_ERR_UNSUPPORTED, "k must be in 1..%d", RTK_KMAX);
    // kdtree.h:182-183 throws std::logic_error here
    if (p->k > c->S.n_photons) return fail(RT_ERR_STATE, "k is greater than the number of nodes");
    if (p->photons_requested == 0) return fail(RT_ERR_INVALID, "photons_requested must be > 0 with use_photons");
  }
  return RT_OK;
}

int read_counters(rt_ctx* c, rt_stats* st) {
  unsigned long long h[RTK_CNT_COUNT];
  HIP_TRY(hipMemcpy(h, c->dCounters, sizeof h, hipMemcpyDeviceToHost));
  st->rays_closest = h[RTK_CNT_CLOSEST];
  st->rays_shadow = h[RTK_CNT_SHADOW];
  st->knn_queries = h[RTK_CNT_KNN];
  st->nodes_visited = h[RTK_CNT_NODES];
  st->tris_tested = h[RTK_CNT_TRIS];
  st->kd_visited = h[RTK_CNT_KD];
  st->frame_fetches = h[RTK_CNT_FRAMES];
  st->reserved[0] = h[RTK_CNT_WNODE];  // diagnostics (collect_stats): wave-level node steps,
  st->reserved[1] = h[RTK_CNT_WLEAF];  // wave-level leaf phases -> lane utilisation of the traversal
  st->reserved[2] = h[RTK_CNT_LWAIT];  // lanes holding a leaf / lanes without a ray at the START of the round,
  st->reserved[3] = h[RTK_CNT_LIDLE];  // summed over that round's node steps
#ifdef RT_PHASE_TIMING  // diagnostic build (tools/phase_timing.sh): section clocks of the pooled kernel
  if (getenv("RT_PHASE_DUMP")) {
    fprintf(stderr, "{\"phase_clocks\": [");
    for (int i = 16; i < RTK_CNT_COUNT; ++i) fprintf(stderr, "%llu%s", h[i], i + 1 < RTK_CNT_COUNT ? ", " : "]}\n");
  }
#endif
  return RT_OK;
}

// Launch the integrate kernel for p on `stream`, bracketed by an event pair.
int launch_frame(rt_ctx* c, const rt_params* p, float4* dAccum, hipStream_t stream, int* evIndex) {
  const uint32_t sppCount = p->spp_count ? p->spp_count : p->spp;
  const uint32_t sshift = choose_sshift(c, p, sppCount);
  int rc = ensure_tiles(c, p, sshift);
  if (rc != RT_OK) return rc;
  rtk::RenderArgs A;
  A.sshift = sshift;
  wave_tile_shape(sshift, A.tileW, A.tileH);
  A.tiles = c->dTiles, A.n_tiles = c->nTiles;
  A.width = p->width, A.height = p->height, A.spp = p->spp;
  A.s0 = p->spp_count ? p->spp_begin : 0;
  A.s1 = p->spp_count ? p->spp_begin + p->spp_count : p->spp;
  A.mode = p->mode, A.max_depth = p->max_depth, A.seed = p->seed;
  A.k = p->k, A.photons_requested = p->photons_requested;
  static const bool noPool = getenv("RT_NO_POOL") != nullptr;
  A.flags = (noPool || (p->reserved[1] & 1u)) ? 0u : 1u;
  // stack entries: one per inner level on a root-to-leaf path; the photon k-NN
  // keeps one split distance per kd level in the same region
  // a root-to-leaf path of depth d passes d inner nodes and each stacks at most one
  // far child, so d entries suffice
  uint32_t levels = c->bvh.maxDepth > 1 ? c->bvh.maxDepth : 1;
  A.kd16 = 0;
  if (p->use_photons) {
    uint32_t kd = 1;
    while ((1ull << kd) <= c->S.n_photons) ++kd;
    // the walk's stack: a sentinel + at most one pending far child per tree level + the row written ahead of the top;
    // 16-bit entries (two per word) when every photon index fits
    A.kd16 = c->S.n_photons < 65535u ? 1u : 0u;
    const uint32_t kdRows = A.kd16 ? (kd + 2u + 1u) / 2u : kd + 1u;
    levels = levels > kdRows ? levels : kdRows;
  }
  // (+1: row 0 of a lane's stack is the TERM sentinel, rt_kernels.hip Trav)
  A.stackLevels = (levels > (uint32_t)rtbvh::kMaxDepth ? (uint32_t)rtbvh::kMaxDepth : levels) + 1u;
  A.tileCounter = c->dTileCounter, A.numCUs = c->numCUs, A.waveWords = 0, A.tilesPerBlock = 1;
  const int e = c->evUsed % kEventPairs;
  // rt_params.reserved[2] bit 0: the queue-based (wavefront) integrator — BVH direct lighting with
  // at most 3 lights, like the pooled kernel; same frame bit for bit
  if ((p->reserved[2] & 1u) && !p->use_photons && p->accel != RT_ACCEL_BRUTE && c->S.n_lights <= 3u) {
    rt_ctx::GranList G;
    rc = ensure_granules(c, p, p->rank, &G);
    if (rc != RT_OK) return rc;
    const size_t perSample = (size_t)G.n * 64u;
    if (perSample == 0) {
      HIP_TRY(hipEventRecord(c->ev[e][0], stream));
      HIP_TRY(hipEventRecord(c->ev[e][1], stream));
      c->evUsed++;
      if (evIndex) *evIndex = e;
      return RT_OK;
    }
    size_t batch = (4u << 20) / perSample;  // ~4 M paths per batch (1.1 GB of state + queues)
    batch = batch < 1 ? 1 : batch > sppCount ? sppCount : batch;
    const size_t P = batch * perSample;
    if (P > c->wfCap) {
      if (c->wfBlock) HIP_TRY(hipFree(c->wfBlock));
      c->wfBlock = nullptr, c->wfCap = 0;
      // rng 4, org 16, dir 16, key 8, nrm 16, pnt 16, col 48, rayO 64, rayD 64, res 32 = 284 B per path
      const size_t bytes = P * 284 + 4096 + 2048 * sizeof(unsigned long long);
      HIP_TRY(hipMalloc(&c->wfBlock, bytes));
      char* q = static_cast<char*>(c->wfBlock);
      auto take = [&](size_t n) {
        char* r = q;
        q += (n + 255) / 256 * 256;
        return r;
      };
      rtk::WfArgs& W = c->wf;
      W.rayO = reinterpret_cast<float4*>(take(P * 64)), W.rayD = reinterpret_cast<float4*>(take(P * 64));
      W.col = reinterpret_cast<float4*>(take(P * 48));
      W.org = reinterpret_cast<float4*>(take(P * 16)), W.dir = reinterpret_cast<float4*>(take(P * 16));
      W.nrm = reinterpret_cast<float4*>(take(P * 16)), W.pnt = reinterpret_cast<float4*>(take(P * 16));
      W.res = reinterpret_cast<uint2*>(take(P * 32)), W.key = reinterpret_cast<uint2*>(take(P * 8));
      W.rng = reinterpret_cast<uint32_t*>(take(P * 4));
      W.stripes = reinterpret_cast<unsigned long long*>(take(2048 * sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(W.stripes, 0, 2048 * sizeof(unsigned long long), stream));
      c->wfCap = P;
    }
    rtk::WfArgs W = c->wf;
    W.gran = G.d, W.nGran = G.n, W.width = p->width, W.height = p->height, W.spp = p->spp, W.seed = p->seed;
    W.s0 = A.s0, W.s1 = A.s1, W.batch = (uint32_t)batch, W.nPaths = 0;
    HIP_TRY(hipEventRecord(c->ev[e][0], stream));
    hipError_t hw = rtk::launch_wavefront(c->S, W, p->mode, p->max_depth, dAccum, c->dCounters, c->dTileCounter, A.stackLevels, c->numCUs, stream);
    if (hw != hipSuccess) return fail(RT_ERR_HIP, "wavefront launch failed: %s", hipGetErrorString(hw));
    HIP_TRY(hipEventRecord(c->ev[e][1], stream));
    c->evUsed++;
    if (evIndex) *evIndex = e;
    return RT_OK;
  }
  HIP_TRY(hipEventRecord(c->ev[e][0], stream));
  hipError_t he = rtk::launch_render(p->accel == RT_ACCEL_BRUTE, p->use_photons != 0, p->collect_stats != 0, c->S, A,
                                     dAccum, c->dCounters, stream);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "render launch failed: %s", hipGetErrorString(he));
  HIP_TRY(hipEventRecord(c->ev[e][1], stream));
  c->evUsed++;
  if (evIndex) *evIndex = e;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

}  // extern "C"

namespace {
// rt_create with an optional host-built tree to copy instead of building one (rt_group: every
// device gets the same tree, built once)
int create_ctx(const rt_scene_desc* sc, const rt_options* opt, const rtbvh::Built* prebuilt, rt_ctx** out);
}  // namespace

extern "C" {

int rt_create(const rt_scene_desc* sc, const rt_options* opt, rt_ctx** out) { return create_ctx(sc, opt, nullptr, out); }

}  // extern "C"

namespace {
// f(thread, begin, end) over contiguous chunks of [b, e): up to 16 threads, one per 2^18 elements
template <class F>
void par_chunks(size_t b, size_t e, F f) {
  static const uint32_t hw = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  const size_t n = e > b ? e - b : 0;
  const uint32_t T = (uint32_t)std::max<size_t>(1, std::min<size_t>(hw, n >> 18));
  if (T <= 1u) {
    f(0u, b, e);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(T - 1u);
  for (uint32_t t = 1; t < T; ++t) th.emplace_back([&f, b, n, t, T] { f(t, b + n * t / T, b + n * (t + 1) / T); });
  f(0u, b, b + n / T);
  for (std::thread& x : th) x.join();
}

int create_ctx(const rt_scene_desc* sc, const rt_options* opt, const rtbvh::Built* prebuilt, rt_ctx** out) {
  if (!sc || !out) return fail(RT_ERR_INVALID, "scene/out is null");
  *out = nullptr;
  if (!sc->vertex_pos || !sc->vertex_nrm || !sc->tri_vtx || !sc->mesh_tri_begin || !sc->mesh_vtx_begin ||
      !sc->materials || (sc->n_lights && !sc->lights))
    return fail(RT_ERR_INVALID, "scene descriptor has null arrays");
  if (sc->n_meshes == 0 || sc->n_vertices == 0 || sc->n_triangles == 0)
    return fail(RT_ERR_INVALID, "empty scene");
  // (node and leaf refs are 31-bit byte offsets of 32-B node / 48-B triangle records)
  if (sc->n_triangles >= (1u << 25)) return fail(RT_ERR_UNSUPPORTED, "more than 2^25 - 1 triangles");
  int rc = select_device(opt ? opt->device : 0);
  if (rc != RT_OK) return rc;

  rt_ctx* c = new rt_ctx();
  c->device = opt ? opt->device : 0;
  // the tree: host SAH builder, or the device builder (tiny scenes always take the host's
  // special cases)
  // (RT_BVH_GPU=1 / 2 / 3: the device / hybrid / host builder whatever the options say — the test suites run whole on each)
  const char* gpuEnv = getenv("RT_BVH_GPU");
  uint32_t wantBuilder = gpuEnv ? (uint32_t)atoi(gpuEnv) : (opt ? opt->bvh_builder : (uint32_t)RT_BVH_AUTO);
  if (wantBuilder > RT_BVH_HOST) {
    delete c;
    return fail(RT_ERR_INVALID, "unknown bvh_builder %u", wantBuilder);
  }
  // AUTO: the device builder gives the host builder's tree (tests/treedigest.py; profiles/r04_builders.txt) 2 ... 14 x sooner,
  // so every scene it is faster on takes it (from 8,192 triangles: below that a build is 1-3 ms either way and the host needs no
  // device round trip); a group of contexts given a host-built tree shares it
  static const uint32_t autoFrom = getenv("RT_BVH_AUTO_FROM") ? (uint32_t)atoi(getenv("RT_BVH_AUTO_FROM")) : 8192u;
  if (wantBuilder == RT_BVH_AUTO) wantBuilder = (sc->n_triangles >= autoFrom && !prebuilt) ? (uint32_t)RT_BVH_DEVICE : (uint32_t)RT_BVH_HOST;
  // (a scene of a single part has no top to build on the host: the device builder's own path handles it)
  const bool hybrid = wantBuilder == RT_BVH_HYBRID && sc->n_triangles > 1024u;
  const bool gpuBuild = (wantBuilder == RT_BVH_DEVICE || wantBuilder == RT_BVH_HYBRID) && sc->n_triangles >= 16;
  rtbvh::TopBuilt topBuilt;
  rtbvh::ScenePlan plan;
  std::vector<float> sizeKey;
  const auto tBuild0 = std::chrono::steady_clock::now();
  try {
    if (hybrid) {
      rtbvh::buildTop(*sc, opt ? opt->bvh_leaf_max : 0, 1024u, topBuilt);
      if (getenv("RT_BVH_VERBOSE"))
        fprintf(stderr, "hybrid builder: host top of %zu nodes over %zu parts in %.1f ms\n", topBuilt.nodes.size(), topBuilt.parts.size(),
                std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tBuild0).count());
      c->bvh.leafMax = topBuilt.leafMax, c->bvh.pad = topBuilt.pad, c->bvh.originBound = topBuilt.originBound, c->bvh.boxScale = topBuilt.boxScale;
      c->bvh.depthCap = topBuilt.depthCap;
      (void)rtbvh::planSceneExact(*sc, opt ? opt->bvh_leaf_max : 0, sizeKey);  // (the size keys of the subtrees' sweeps)
    } else if (gpuBuild) {
      // (the device build restates the host builder's splits: it takes the host's depth cap and size keys)
      plan = rtbvh::planSceneExact(*sc, opt ? opt->bvh_leaf_max : 0, sizeKey);
      if (getenv("RT_BVH_VERBOSE"))
        fprintf(stderr, "device builder: validation + size keys in %.2f ms\n", std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tBuild0).count());
      c->bvh.leafMax = plan.leafMax, c->bvh.pad = plan.pad, c->bvh.originBound = plan.originBound, c->bvh.boxScale = plan.boxScale;
      c->bvh.depthCap = plan.depthCap;
    } else if (prebuilt) {
      c->bvh = *prebuilt;
    } else {
      rtbvh::build(*sc, opt ? opt->bvh_leaf_max : 0, c->bvh);
    }
  } catch (const std::exception& e) {
    delete c;
    return fail(RT_ERR_INVALID, "scene rejected: %s", e.what());
  }
  static_assert(sizeof(rtbvh::Node16) == 2 * sizeof(uint4), "node layout");
  static_assert(sizeof(rtbvh::TriRec) == 3 * sizeof(float4), "triangle layout");
  // (big scenes: the per-triangle and per-vertex host passes of rt_create are shared by a few threads — 8 M triangles
  // spent 60 ms in them on one)
  std::unique_ptr<uint4[]> shade(new uint4[sc->n_triangles]);
  for (uint32_t m = 0; m < sc->n_meshes; ++m)
    par_chunks(sc->mesh_tri_begin[m], sc->mesh_tri_begin[m + 1], [&](uint32_t, size_t tb, size_t te) {
      for (size_t t = tb; t < te; ++t) shade[t] = make_uint4(sc->tri_vtx[3 * t], sc->tri_vtx[3 * t + 1], sc->tri_vtx[3 * t + 2], m);
    });

  rtk::DevScene& S = c->S;
#define UP(field, src, n)                                  \
  if ((rc = upload_owned(c, &S.field, src, n)) != RT_OK) { \
    rt_destroy(c);                                         \
    return rc;                                             \
  }
  UP(triShade, shade.get(), (size_t)sc->n_triangles);
  shade.reset();
  UP(vpos, sc->vertex_pos, (size_t)sc->n_vertices * 3);
  UP(vnrm, sc->vertex_nrm, (size_t)sc->n_vertices * 3);
  if (gpuBuild) {
    if (getenv("RT_BVH_VERBOSE"))
      fprintf(stderr, "scene arrays on the device %.2f ms after the start\n", std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tBuild0).count());
    rtk::GpuBvh G;
    hipError_t he = hipSuccess;
    if (hybrid) {
      he = rtk::gpu_bvh_build_over_top(S.vpos, S.triShade, sizeKey.data(), sc->n_triangles, topBuilt, &G, nullptr);
    } else {
      he = rtk::gpu_bvh_build_exact(S.vpos, S.triShade, sizeKey.data(), sc->n_triangles, plan, &G, nullptr);
    }
    if (he != hipSuccess) {
      rt_destroy(c);
      return fail(RT_ERR_HIP, "device BVH build failed: %s", hipGetErrorString(he));
    }
    S.nodes = G.nodes16, S.tris = G.tris, S.trisRef = G.trisRef;
    c->allocs.push_back(G.nodes16), c->allocs.push_back(G.tris), c->allocs.push_back(G.trisRef);
    c->dNodesF = G.nodesF;
    c->bvh.maxDepth = G.maxDepth;
    S.n_nodes = G.n_nodes;
    c->builder = hybrid ? RT_BVH_HYBRID : RT_BVH_DEVICE;
    // Trees whose top the render kernel may keep in LDS (rt_kernels.hip plan_persist: a prefix of the node array) get the host
    // builder's final numbering — the most-visited nodes first, greedily by box area from the root (bvh_build.cpp
    // relayoutTop) — instead of the device's pre-order: C4 loses 2 % on a pre-order tree.  64 KB ... 4 MB back and forth.
    if (S.n_nodes >= 2u && S.n_nodes <= 65536u) {
      try {
        c->bvh.nodes.resize(S.n_nodes);
        if (hipMemcpy(c->bvh.nodes.data(), c->dNodesF, (size_t)S.n_nodes * sizeof(rtbvh::Node), hipMemcpyDeviceToHost) != hipSuccess)
          throw std::runtime_error("reading the device-built tree back failed");
        rtbvh::relayoutAndPack(c->bvh);
        if (hipMemcpy(c->dNodesF, c->bvh.nodes.data(), (size_t)S.n_nodes * sizeof(rtbvh::Node), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(const_cast<uint4*>(S.nodes), c->bvh.nodes16.data(), (size_t)S.n_nodes * sizeof(rtbvh::Node16), hipMemcpyHostToDevice) != hipSuccess)
          throw std::runtime_error("writing the renumbered tree failed");
        c->bvh.nodes.clear(), c->bvh.nodes16.clear();  // (rt_bvh_export reads the device copies)
      } catch (const std::exception& e) {
        rt_destroy(c);
        return fail(RT_ERR_HIP, "device BVH build: %s", e.what());
      }
    }
  } else {
    UP(nodes, c->bvh.nodes16.data(), c->bvh.nodes16.size() * 2);
    UP(tris, c->bvh.tris.data(), c->bvh.tris.size() * 3);
    UP(trisRef, c->bvh.trisRef.data(), c->bvh.trisRef.size() * 3);
    S.n_nodes = static_cast<uint32_t>(c->bvh.nodes.size());
  }
  // The node records the pooled render kernel and rt_trace traverse (rt_options.node_format; RT_NODES=f16|q8 overrides).
  // RT_NODES_Q8 — 16-byte records, ONE vector-memory request per visit (bvh_build.h Slot16) — is for trees the caches do
  // not hold, where the traversal sits on the vector L1's request rate.  The other kernels (photon emission, ray streams,
  // the wavefront integrator, the one-wave-per-workgroup render instances) keep the 32-byte records, so both forms are resident.
  {
    uint32_t want = opt ? opt->node_format : (uint32_t)RT_NODES_AUTO;
    if (const char* e = getenv("RT_NODES")) want = !strcmp(e, "q8") ? (uint32_t)RT_NODES_Q8 : !strcmp(e, "f16") ? (uint32_t)RT_NODES_F16 : want;
    if (want > RT_NODES_Q8) {
      rt_destroy(c);
      return fail(RT_ERR_INVALID, "unknown node_format %u", want);
    }
    if (want == RT_NODES_AUTO) want = RT_NODES_F16;
    if (want == RT_NODES_Q8) {
      try {
        if (gpuBuild) {  // the packer works from the float records: fetch what the device builder left on the device
          c->bvh.nodes.resize(S.n_nodes), c->bvh.tris.resize(sc->n_triangles), c->bvh.trisRef.resize(sc->n_triangles);
          if (hipMemcpy(c->bvh.nodes.data(), c->dNodesF, (size_t)S.n_nodes * sizeof(rtbvh::Node), hipMemcpyDeviceToHost) != hipSuccess ||
              hipMemcpy(c->bvh.tris.data(), S.tris, (size_t)sc->n_triangles * sizeof(rtbvh::TriRec), hipMemcpyDeviceToHost) != hipSuccess ||
              hipMemcpy(c->bvh.trisRef.data(), S.trisRef, (size_t)sc->n_triangles * sizeof(rtbvh::TriRec), hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("reading the device-built tree back failed");
        }
        if (c->bvh.q8.empty()) rtbvh::packQ8(c->bvh);
        if (gpuBuild) c->bvh.nodes.clear(), c->bvh.tris.clear(), c->bvh.trisRef.clear();  // (rt_bvh_export reads the device copies)
      } catch (const std::exception& e) {
        rt_destroy(c);
        return fail(RT_ERR_UNSUPPORTED, "node_format RT_NODES_Q8: %s", e.what());
      }
      static_assert(sizeof(rtbvh::Slot16) == sizeof(uint4), "slot layout");
      if ((rc = upload_owned(c, &S.q8, c->bvh.q8.data(), c->bvh.q8.size())) != RT_OK) {
        rt_destroy(c);
        return rc;
      }
      S.q8ShiftBytes = c->bvh.q8Shift + 4u;
      c->nodeFormat = RT_NODES_Q8;
      std::vector<rtbvh::Slot16>().swap(c->bvh.q8);  // (the host copy is not needed again)
    }
  }
  c->buildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tBuild0).count();
  if (getenv("RT_BVH_VERBOSE")) fprintf(stderr, "tree resident %.2f ms after the start\n", c->buildMs);
  UP(mats, sc->materials, sc->n_meshes);
  {
    std::vector<rtd::DevMat> dm(sc->n_meshes);
    for (uint32_t m = 0; m < sc->n_meshes; ++m) dm[m] = rtd::make_dev_mat(sc->materials[m]);
    UP(matsDev, dm.data(), dm.size());
  }
  UP(lights, sc->lights, sc->n_lights);
  UP(meshTriBegin, sc->mesh_tri_begin, sc->n_meshes + 1);
  UP(meshVtxBegin, sc->mesh_vtx_begin, sc->n_meshes + 1);
#undef UP
  S.n_tris = sc->n_triangles;
  S.n_lights = sc->n_lights;
  S.n_photons = 0;
  S.invBoxScale = 1.f / c->bvh.boxScale;
  S.originBound = c->bvh.originBound;
  // Pool thresholds (Trav::round's descent early exit, the steal and refill levels).  Two scene
  // classes, as for the samples-of-a-pixel-per-wave rule: trees the caches hold (<= 65,536 nodes)
  // are issue-bound and want long descents (12 / 8 / 24: C2 50.5 ms; 16 or 24 lanes cost 0.2-1 %);
  // beyond that every step waits on the vector L1, and leaving the descent with up to 24 lanes still
  // in it plus refilling at 32 hands out work sooner (C5 328.5 -> 317.4 ms, C5x8 55.0 -> 52.4 ms;
  // profiles/r03_pool_thresholds.txt).
  // The short reciprocal (rt_device.h recip_fast: v_rcp_f32 + one Newton step, the division's bits for 2^-100 <= |x| < 2^101
  // — exhaustive check, tools/microbench/recip_exact.hip) replaces the division in the default render instances
  // (rt_kernels.hip LT_FASTDET) in two places, and the host vouches for the range here:
  //  * 1 / det of the triangle test: |det| = |e1 . (d x e2)| <= |e1| |e2| |d|; edges are at most 2 sqrt(3) maxAbs long, and
  //    the rays the RENDER kernels make are unit vectors (camera, bounce) or run from a surface point to a light sample;
  //  * length and 1 / length in normalisations (sqrt_fast: the same check, 2^-100 <= x < 2^101): every vector the kernels
  //    normalise is a small sum of scene inputs, so |input| <= 1e14 keeps the squared length below 2^100 (the lower end
  //    is tested per lane: rt_device.h unit3).
  // Outside these bounds — or with any non-finite input — the kernels divide.  (User rays, rt_trace /
  // rt_trace_stream_device, always divide.)
  {
    double maxAbs = 0, maxLight = 0, maxAny = 0;
    bool finite = true;
    // (a non-finite value has all exponent bits set: its magnitude bits compare above every finite float's)
    auto eat = [&](const float* p, size_t n) {
      uint32_t top[64] = {0};
      par_chunks(0, n, [&](uint32_t th, size_t b, size_t e) {
        uint32_t m = 0;
        for (size_t i = b; i < e; ++i) {
          uint32_t u;
          memcpy(&u, p + i, 4);
          u &= 0x7fffffffu;
          m = u > m ? u : m;
        }
        top[th] = m;
      });
      uint32_t m = 0;
      for (uint32_t t : top) m = t > m ? t : m;
      float f;
      memcpy(&f, &m, 4);
      if (m >= 0x7f800000u) finite = false;
      else maxAny = std::max(maxAny, (double)f);
      return m >= 0x7f800000u ? 0.0 : (double)f;
    };
    maxAbs = eat(sc->vertex_pos, 3 * (size_t)sc->n_vertices);
    eat(sc->vertex_nrm, 3 * (size_t)sc->n_vertices);
    eat(sc->camera.position, 12);
    for (uint32_t l = 0; l < sc->n_lights; ++l) {
      const rt_light& L = sc->lights[l];
      eat(L.position, 15);
      eat(&L.intensity, 6);
      double pos = 0, ver = 0, hor = 0;
      for (int a = 0; a < 3; ++a) pos += (double)L.position[a] * L.position[a], ver += (double)L.vertical[a] * L.vertical[a], hor += (double)L.horizontal[a] * L.horizontal[a];
      maxLight = std::max(maxLight, std::sqrt(pos) + std::fabs((double)L.side) * (std::sqrt(ver) + std::sqrt(hor)));
    }
    const double edge = 2.0 * 1.7320508 * maxAbs, dir = 1.7320508 * maxAbs + maxLight + 2.0;
    const double detBound = 1.01 * edge * edge * dir;
    S.slowRecip = (finite && maxAny <= 1e14 && maxLight <= 1e14 && std::isfinite(detBound) && detBound < 1.2676506e30) ? 0u : 1u;
    if (getenv("RT_SLOW_RECIP")) S.slowRecip = 1u;  // (A/B and the parity tests of the division path)
    // ... and the device vouches for the short forms itself, once per process and device (2^25 inputs, well under a
    // millisecond): the exhaustive check ran on one MI355X; a part whose v_rcp_f32 / v_rsq_f32 rounded differently would
    // show here, and its contexts divide.
    if (!S.slowRecip) {
      static std::mutex mu;
      static std::map<int, bool> verified;
      std::lock_guard<std::mutex> lock(mu);
      auto it = verified.find(c->device);
      if (it == verified.end()) {
        bool ok = false;
        uint32_t* dBad = nullptr;
        uint32_t bad = 1u;
        if (hipMalloc(reinterpret_cast<void**>(&dBad), sizeof(uint32_t)) == hipSuccess) {
          if (hipMemset(dBad, 0, sizeof(uint32_t)) == hipSuccess && rtk::launch_selfcheck_recip(dBad, nullptr) == hipSuccess &&
              hipMemcpy(&bad, dBad, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess)
            ok = bad == 0u;
          (void)hipFree(dBad);
        }
        it = verified.emplace(c->device, ok).first;
        if (getenv("RT_BVH_VERBOSE")) fprintf(stderr, "short reciprocal / square root self-check on device %d: %s\n", c->device, ok ? "bit-identical" : "MISMATCH, dividing");
      }
      if (!it->second) S.slowRecip = 1u;
      c->recipCheck = it->second ? 1u : 2u;
    }
  }
  const bool bigTree = S.n_nodes > 65536;
  S.leafT = getenv("RT_LEAFT") ? atoi(getenv("RT_LEAFT")) : bigTree ? 32 : 12;
  S.leafMul = getenv("RT_LEAFMUL") ? atoi(getenv("RT_LEAFMUL")) : bigTree ? 32 : 22;
  S.stealT = getenv("RT_STEALT") ? atoi(getenv("RT_STEALT")) : 8;
  S.refillT = getenv("RT_REFILLT") ? atoi(getenv("RT_REFILLT")) : bigTree ? 32 : 24;
  S.phPos = S.phDir = nullptr, S.phTopo = nullptr;
  S.topK = 0;
  S.cam = sc->camera;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->numCUs = (uint32_t)cus;
    if (hipMalloc(reinterpret_cast<void**>(&c->dTileCounter), sizeof(uint32_t)) != hipSuccess) {
      rt_destroy(c);
      return fail(RT_ERR_HIP, "tile counter allocation failed");
    }
    if (c->numCUs == 0) {
      rt_destroy(c);
      return fail(RT_ERR_HIP, "device %d reports no compute units", c->device);
    }
  }
  // (the counter block + 1,024 striped slots x 4 for the one-wave-per-workgroup kernels: rt_kernels.hip flush_stats_striped)
  if (hipMalloc(reinterpret_cast<void**>(&c->dCounters), (RTK_CNT_COUNT + 4096) * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(c->dCounters, 0, (RTK_CNT_COUNT + 4096) * sizeof(unsigned long long)) != hipSuccess) {
    rt_destroy(c);
    return fail(RT_ERR_HIP, "counter allocation failed");
  }
  for (auto& pr : c->ev)
    if (hipEventCreate(&pr[0]) != hipSuccess || hipEventCreate(&pr[1]) != hipSuccess) {
      rt_destroy(c);
      return fail(RT_ERR_HIP, "event creation failed");
    }
  c->evReady = true;
  *out = c;
  return RT_OK;
}
}  // namespace

extern "C" {

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->phPos) (void)hipFree(c->phPos);
  if (c->phDir) (void)hipFree(c->phDir);
  if (c->phTopo) (void)hipFree(c->phTopo);
  if (c->dTiles) (void)hipFree(c->dTiles);
  if (c->dCounters) (void)hipFree(c->dCounters);
  if (c->dTileCounter) (void)hipFree(c->dTileCounter);
  if (c->dNodesF) (void)hipFree(c->dNodesF);
  if (c->wfBlock) (void)hipFree(c->wfBlock);
  for (auto& kv : c->granules)
    if (kv.second.d) (void)hipFree(kv.second.d);
  if (c->evReady)
    for (auto& pr : c->ev) (void)hipEventDestroy(pr[0]), (void)hipEventDestroy(pr[1]);
  delete c;
}

int rt_set_photons(rt_ctx* c, const float* pos3, const float* dir3, uint32_t n) {
  if (!c) return fail(RT_ERR_INVALID, "ctx is null");
  if (n && (!pos3 || !dir3)) return fail(RT_ERR_INVALID, "photon arrays are null");
  if (n >= (1u << 30)) return fail(RT_ERR_UNSUPPORTED, "too many photons");
  HIP_TRY(hipSetDevice(c->device));
  // the device scene forgets the old map BEFORE it is freed: a failed upload must leave
  // "no photons" behind (check_params then refuses use_photons), never dangling pointers
  c->S.phPos = c->S.phDir = nullptr, c->S.phTopo = nullptr, c->S.n_photons = 0;
  float4 *oldP = c->phPos, *oldD = c->phDir;
  uint4* oldT = c->phTopo;
  c->phPos = c->phDir = nullptr, c->phTopo = nullptr;
  if (oldP) HIP_TRY(hipFree(oldP));
  if (oldD) HIP_TRY(hipFree(oldD));
  if (oldT) HIP_TRY(hipFree(oldT));
  std::vector<float4> p(n), d(n);
  for (uint32_t i = 0; i < n; ++i) {
    p[i] = make_float4(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2], 0.f);
    d[i] = make_float4(dir3[3 * (size_t)i], dir3[3 * (size_t)i + 1], dir3[3 * (size_t)i + 2], 0.f);
  }
  int rc = upload(&c->phPos, p.data(), n);
  if (rc == RT_OK) rc = upload(&c->phDir, d.data(), n);
  if (rc == RT_OK && n) {  // the arrays come in tree order (kdtree.h:60-69): the explicit topology follows from it
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&c->phTopo), 2 * (size_t)n * sizeof(uint4));
    if (he == hipSuccess) he = rtk::launch_kd_topology(c->phPos, n, c->phTopo, nullptr);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) rc = fail(RT_ERR_HIP, "photon topology failed: %s", hipGetErrorString(he));
  }
  if (rc != RT_OK) {  // drop whatever part of the map made it
    if (c->phPos) (void)hipFree(c->phPos);
    if (c->phDir) (void)hipFree(c->phDir);
    if (c->phTopo) (void)hipFree(c->phTopo);
    c->phPos = c->phDir = nullptr, c->phTopo = nullptr;
    return rc;
  }
  c->S.phPos = c->phPos, c->S.phDir = c->phDir, c->S.phTopo = c->phTopo, c->S.n_photons = n;
  return RT_OK;
}

int rt_emit_photons(rt_ctx* c, uint32_t n_requested, uint32_t seed, float* pos3, float* dir3, float* weight,
                    uint32_t* n_out) {
  if (!c || !n_out) return fail(RT_ERR_INVALID, "ctx/n_out is null");
  *n_out = 0;
  if (n_requested == 0 || c->S.n_lights == 0) return RT_OK;
  if (!pos3 || !dir3) return fail(RT_ERR_INVALID, "output arrays are null");
  HIP_TRY(hipSetDevice(c->device));
  // PhotonMap.h:19-20: lightPdf = 1.f / #lights; photonsPerLS = (int)(n * lightPdf)
  const float lightPdf = 1.f / static_cast<float>(c->S.n_lights);
  const uint32_t perLight = static_cast<uint32_t>(static_cast<int>(static_cast<float>(static_cast<int>(n_requested)) * lightPdf));
  const uint32_t n = perLight * c->S.n_lights;
  if (n == 0) return RT_OK;
  float4 *dPos = nullptr, *dDir = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dPos), n * sizeof(float4)));
  if (hipMalloc(reinterpret_cast<void**>(&dDir), n * sizeof(float4)) != hipSuccess) {
    (void)hipFree(dPos);
    return fail(RT_ERR_HIP, "photon buffer allocation failed");
  }
  hipError_t he = rtk::launch_emit(c->S, perLight, seed, dPos, dDir, c->dCounters, nullptr);
  std::vector<float4> hp(n), hd(n);
  if (he == hipSuccess) he = hipMemcpy(hp.data(), dPos, n * sizeof(float4), hipMemcpyDeviceToHost);
  if (he == hipSuccess) he = hipMemcpy(hd.data(), dDir, n * sizeof(float4), hipMemcpyDeviceToHost);
  (void)hipFree(dPos), (void)hipFree(dDir);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "photon emission failed: %s", hipGetErrorString(he));
  uint32_t m = 0;
  for (uint32_t j = 0; j < n; ++j) {
    if (hp[j].w == 0.f) continue;  // this emitted photon stored no particle
    pos3[3 * (size_t)m] = hp[j].x, pos3[3 * (size_t)m + 1] = hp[j].y, pos3[3 * (size_t)m + 2] = hp[j].z;
    dir3[3 * (size_t)m] = hd[j].x, dir3[3 * (size_t)m + 1] = hd[j].y, dir3[3 * (size_t)m + 2] = hd[j].z;
    if (weight) weight[m] = hd[j].w;
    ++m;
  }
  *n_out = m;
  return RT_OK;
}

int rt_render_device(rt_ctx* c, const rt_params* p, void* d_accum, void* stream, rt_stats* stats) {
  if (!c || !d_accum) return fail(RT_ERR_INVALID, "ctx/d_accum is null");
  int rc = check_params(c, p);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stats) HIP_TRY(hipMemsetAsync(c->dCounters, 0, RTK_CNT_COUNT * sizeof(unsigned long long), s));
  int e = 0;
  rc = launch_frame(c, p, static_cast<float4*>(d_accum), s, &e);
  if (rc != RT_OK) return rc;
  if (stats) {
    memset(stats, 0, sizeof *stats);
    HIP_TRY(hipStreamSynchronize(s));
    rc = read_counters(c, stats);
    if (rc != RT_OK) return rc;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[e][0], c->ev[e][1]));
    stats->kernel_ms = ms;
    const uint32_t n = p->spp_count ? p->spp_count : p->spp;
    // samples of the tiles this rank owns
    stats->samples = 0;
    if (c->nTiles) {
      uint64_t px = 0;
      const uint32_t tile = c->tileKey.tile, world = c->tileKey.world;
      for (uint32_t y = 0; y < p->height; ++y)
        for (uint32_t x = 0; x < p->width; ++x)
          if (world <= 1 || ((x / tile) + (y / tile)) % world == p->rank) ++px;
      stats->samples = px * n;
    }
  }
  return RT_OK;
}

int rt_resolve_device(rt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, const void* d_accum,
                      const void* d_bg, void* d_out, void* stream) {
  if (!c || !d_accum || !d_bg || !d_out) return fail(RT_ERR_INVALID, "null argument");
  if (spp == 0) return fail(RT_ERR_INVALID, "spp must be >= 1");
  HIP_TRY(hipSetDevice(c->device));
  hipError_t he = rtk::launch_resolve(width * height, spp, static_cast<const float4*>(d_accum),
                                      static_cast<const float*>(d_bg), static_cast<float*>(d_out),
                                      static_cast<hipStream_t>(stream));
  if (he != hipSuccess) return fail(RT_ERR_HIP, "resolve launch failed: %s", hipGetErrorString(he));
  return RT_OK;
}

int rt_render(rt_ctx* c, const rt_params* p, const float* bg, float* out_rgb, float* accum_out, rt_stats* stats) {
  if (!c) return fail(RT_ERR_INVALID, "ctx is null");
  int rc = check_params(c, p);
  if (rc != RT_OK) return rc;
  if (out_rgb && !bg) return fail(RT_ERR_INVALID, "out_rgb requested without a background image");
  HIP_TRY(hipSetDevice(c->device));
  const size_t npx = (size_t)p->width * p->height;
  float4* dAccum = nullptr;
  float *dBg = nullptr, *dOut = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dAccum), npx * sizeof(float4)));
  auto cleanup = [&]() {
    (void)hipFree(dAccum);
    if (dBg) (void)hipFree(dBg);
    if (dOut) (void)hipFree(dOut);
  };
  hipError_t he = hipMemset(dAccum, 0, npx * sizeof(float4));
  if (he != hipSuccess) {
    cleanup();
    return fail(RT_ERR_HIP, "memset failed: %s", hipGetErrorString(he));
  }
  rt_stats local;
  rc = rt_render_device(c, p, dAccum, nullptr, stats ? stats : &local);
  if (rc != RT_OK) {
    cleanup();
    return rc;
  }
  if (out_rgb) {
    if ((he = hipMalloc(reinterpret_cast<void**>(&dBg), npx * 3 * sizeof(float))) == hipSuccess &&
        (he = hipMalloc(reinterpret_cast<void**>(&dOut), npx * 3 * sizeof(float))) == hipSuccess &&
        (he = hipMemcpy(dBg, bg, npx * 3 * sizeof(float), hipMemcpyHostToDevice)) == hipSuccess) {
      rc = rt_resolve_device(c, p->width, p->height, p->spp, dAccum, dBg, dOut, nullptr);
      if (rc == RT_OK) he = hipMemcpy(out_rgb, dOut, npx * 3 * sizeof(float), hipMemcpyDeviceToHost);
    }
    if (rc == RT_OK && he != hipSuccess) rc = fail(RT_ERR_HIP, "resolve failed: %s", hipGetErrorString(he));
  }
  if (rc == RT_OK && accum_out) {
    he = hipMemcpy(accum_out, dAccum, npx * sizeof(float4), hipMemcpyDeviceToHost);
    if (he != hipSuccess) rc = fail(RT_ERR_HIP, "accumulator read-back failed: %s", hipGetErrorString(he));
  }
  cleanup();
  return rc;
}

int rt_render_passes(rt_ctx* c, const rt_params* p, const float* bg, float* accum_io, float* out_rgb, rt_stats* stats) {
  if (!c || !accum_io || !bg || !out_rgb) return fail(RT_ERR_INVALID, "null argument");
  int rc = check_params(c, p);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(c->device));
  const size_t npx = (size_t)p->width * p->height;
  const uint32_t soFar = p->spp_count ? p->spp_begin + p->spp_count : p->spp;
  float4* dAccum = nullptr;
  float *dBg = nullptr, *dOut = nullptr;
  hipError_t he = hipMalloc(reinterpret_cast<void**>(&dAccum), npx * sizeof(float4));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dBg), npx * 3 * sizeof(float));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dOut), npx * 3 * sizeof(float));
  if (he == hipSuccess) he = hipMemcpy(dAccum, accum_io, npx * sizeof(float4), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(dBg, bg, npx * 3 * sizeof(float), hipMemcpyHostToDevice);
  rt_stats local;
  if (he == hipSuccess) rc = rt_render_device(c, p, dAccum, nullptr, stats ? stats : &local);
  if (he == hipSuccess && rc == RT_OK) rc = rt_resolve_device(c, p->width, p->height, soFar, dAccum, dBg, dOut, nullptr);
  if (he == hipSuccess && rc == RT_OK) he = hipMemcpy(accum_io, dAccum, npx * sizeof(float4), hipMemcpyDeviceToHost);
  if (he == hipSuccess && rc == RT_OK) he = hipMemcpy(out_rgb, dOut, npx * 3 * sizeof(float), hipMemcpyDeviceToHost);
  if (dAccum) (void)hipFree(dAccum);
  if (dBg) (void)hipFree(dBg);
  if (dOut) (void)hipFree(dOut);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "progressive render failed: %s", hipGetErrorString(he));
  return rc;
}

int rt_trace(rt_ctx* c, const rt_ray* rays, uint32_t n, uint32_t accel, uint32_t kind, rt_hit* hits) {
  if (!c || (n && (!rays || !hits))) return fail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(c->device));
  rt_ray* dR = nullptr;
  rt_hit* dH = nullptr;
  int rc = upload(&dR, rays, n);
  if (rc != RT_OK) return rc;
  if (hipMalloc(reinterpret_cast<void**>(&dH), n * sizeof(rt_hit)) != hipSuccess) {
    (void)hipFree(dR);
    return fail(RT_ERR_HIP, "hit buffer allocation failed");
  }
  rtk::DevScene Su = c->S;
  Su.slowRecip = 1u;  // the caller's rays: any length
  hipError_t he = rtk::launch_trace(accel == RT_ACCEL_BRUTE, kind == RT_TRACE_ANY, Su, dR, n, dH, c->dCounters, nullptr);
  if (he == hipSuccess) he = hipMemcpy(hits, dH, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
  (void)hipFree(dR), (void)hipFree(dH);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "trace failed: %s", hipGetErrorString(he));
  return RT_OK;
}

int rt_knn(rt_ctx* c, const float* q3, uint32_t n, uint32_t k, uint32_t* idx, float* dist, uint32_t* visited) {
  if (!c || (n && (!q3 || !idx || !dist))) return fail(RT_ERR_INVALID, "null argument");
  if (c->S.n_photons == 0) return fail(RT_ERR_STATE, "tree is empty");              // kdtree.h:181
  if (k < 1 || k > RTK_KMAX) return fail(RT_ERR_UNSUPPORTED, "k must be in 1..%d", RTK_KMAX);
  if (k > c->S.n_photons) return fail(RT_ERR_STATE, "k is greater than the number of nodes");  // kdtree.h:182-183
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(c->device));
  float* dQ = nullptr;
  uint32_t *dI = nullptr, *dV = nullptr;
  float* dD = nullptr;
  int rc = upload(&dQ, q3, (size_t)n * 3);
  if (rc != RT_OK) return rc;
  hipError_t he = hipMalloc(reinterpret_cast<void**>(&dI), (size_t)n * k * sizeof(uint32_t));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dD), (size_t)n * k * sizeof(float));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dV), (size_t)n * sizeof(uint32_t));
  if (he == hipSuccess) he = rtk::launch_knn(c->S, dQ, n, k, dI, dD, dV, nullptr);
  if (he == hipSuccess) he = hipMemcpy(idx, dI, (size_t)n * k * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (he == hipSuccess) he = hipMemcpy(dist, dD, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost);
  if (he == hipSuccess && visited) he = hipMemcpy(visited, dV, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(dQ);
  if (dI) (void)hipFree(dI);
  if (dD) (void)hipFree(dD);
  if (dV) (void)hipFree(dV);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "knn failed: %s", hipGetErrorString(he));
  return RT_OK;
}

int rt_bvh_info_get(rt_ctx* c, rt_bvh_info* out) {
  if (!c || !out) return fail(RT_ERR_INVALID, "null argument");
  memset(out, 0, sizeof *out);
  out->n_nodes = c->S.n_nodes;
  out->n_tri_records = c->S.n_tris;
  out->max_depth = c->bvh.maxDepth;
  out->leaf_max = c->bvh.leafMax;
  out->pad = c->bvh.pad;
  out->build_ms = c->buildMs;
  out->builder = c->builder;
  out->node_format = c->nodeFormat;
  out->flags = (c->S.slowRecip ? 0u : (uint32_t)RT_BVH_FLAG_SHORT_RECIP) | (c->recipCheck == 2u ? (uint32_t)RT_BVH_FLAG_RECIP_CHECK_FAILED : 0u);
  return RT_OK;
}

int rt_bvh_export(rt_ctx* c, void* nodes64, void* tris48) {
  if (!c) return fail(RT_ERR_INVALID, "ctx is null");
  if (c->builder != RT_BVH_HOST) {  // the arrays only exist on the device
    HIP_TRY(hipSetDevice(c->device));
    if (nodes64) HIP_TRY(hipMemcpy(nodes64, c->dNodesF, (size_t)c->S.n_nodes * sizeof(rtbvh::Node), hipMemcpyDeviceToHost));
    if (tris48) HIP_TRY(hipMemcpy(tris48, c->S.tris, (size_t)c->S.n_tris * sizeof(rtbvh::TriRec), hipMemcpyDeviceToHost));
    return RT_OK;
  }
  if (nodes64) memcpy(nodes64, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(rtbvh::Node));
  if (tris48) memcpy(tris48, c->bvh.tris.data(), c->bvh.tris.size() * sizeof(rtbvh::TriRec));
  return RT_OK;
}

int rt_bvh_build_host(const rt_scene_desc* sc, uint32_t leaf_max, uint32_t threads, rt_bvh_info* info,
                      uint64_t* digest, double* seconds) {
  if (!sc || !info || !digest) return fail(RT_ERR_INVALID, "null argument");
  try {
    rtbvh::Built b;
    const auto t0 = std::chrono::steady_clock::now();
    rtbvh::build(*sc, leaf_max, b, threads);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info->n_nodes = (uint32_t)b.nodes.size(), info->n_tri_records = (uint32_t)b.tris.size();
    info->max_depth = b.maxDepth, info->leaf_max = b.leafMax, info->pad = b.pad;
    uint64_t h = 1469598103934665603ull;  // FNV-1a over float nodes, packed nodes, triangle records
    auto eat = [&h](const void* p, size_t n) {
      const unsigned char* q = static_cast<const unsigned char*>(p);
      for (size_t i = 0; i < n; ++i) h = (h ^ q[i]) * 1099511628211ull;
    };
    eat(b.nodes.data(), b.nodes.size() * sizeof(rtbvh::Node));
    eat(b.nodes16.data(), b.nodes16.size() * sizeof(rtbvh::Node16));
    eat(b.tris.data(), b.tris.size() * sizeof(rtbvh::TriRec));
    *digest = h;
  } catch (const std::exception& e) {
    return fail(RT_ERR_INVALID, "BVH build failed: %s", e.what());
  }
  return RT_OK;
}

int rt_bvh_check_host(const rt_scene_desc* sc, uint32_t leaf_max, uint32_t node_format, uint32_t* out8, double* est2) {
  if (!sc || !out8) return fail(RT_ERR_INVALID, "null argument");
  if (node_format != RT_NODES_F16 && node_format != RT_NODES_Q8) return fail(RT_ERR_INVALID, "node_format must be RT_NODES_F16 or RT_NODES_Q8");
  try {
    rtbvh::Built b;
    rtbvh::build(*sc, leaf_max, b, 0);
    const bool q8 = node_format == RT_NODES_Q8;
    if (q8) rtbvh::packQ8(b);
    const size_t nn = b.nodes16.size();
    if (nn == 0 || nn != b.nodes.size()) return fail(RT_ERR_STATE, "the build produced no packed nodes");
    const float inv = 1.f / b.boxScale;
    std::vector<uint32_t> seen(b.tris.size(), 0);
    std::vector<uint8_t> visited(q8 ? b.q8.size() : nn, 0);
    uint32_t depthSeen = 0;
    double areaSum = 0;  // inner children's decoded box areas (the visit estimate of the packed form)
    // depth-first over the packed tree; returns the float box of the geometry below a child ref
    struct Bx { float lo[3], hi[3]; };
    struct Walker {
      const rt_scene_desc& sc; const rtbvh::Built& b; std::vector<uint32_t>& seen; std::vector<uint8_t>& visited;
      uint32_t& depthSeen; double& areaSum; float inv; bool q8; uint64_t nodes; std::string err;
      Bx walk(int32_t ref, uint32_t depth) {
        Bx r;
        for (int a = 0; a < 3; ++a) r.lo[a] = 3e38f, r.hi[a] = -3e38f;
        depthSeen = std::max(depthSeen, depth);
        if (ref < 0) {  // leaf: ~(byte offset of the first record | count - 1)
          const uint32_t code = ~(uint32_t)ref, cnt = (code & 7u) + 1u;
          if (cnt > b.leafMax) err = "leaf holds more records than leaf_max";
          for (uint32_t t = 0; t < cnt && err.empty(); ++t) {
            rtbvh::TriRec rec;
            if (q8) {  // the records sit in the unified array: identify them by their id, compare with the leaf-order array
              const size_t slot = ((size_t)(code & ~7u) >> 4) + 3u * t;
              if (((code & ~7u) & 15u) || slot + 3 > b.q8.size()) { err = "leaf offset outside the unified array"; break; }
              for (int k = 0; k < 3; ++k)
                if (visited[slot + k]++) err = "a triangle slot is referenced twice";
              memcpy(&rec, &b.q8[slot], sizeof rec);
              if (rec.id >= seen.size() || memcmp(&rec, &b.trisRef[rec.id], sizeof rec) != 0) { err = "a triangle record of the unified array is not the scene's"; break; }
              seen[rec.id]++;
            } else {
              const uint32_t i = (code & ~7u) / 48u + t;
              if ((code & ~7u) % 48u) err = "leaf offset is not a multiple of 48";
              if (i >= seen.size()) { err = "leaf range beyond the triangle array"; break; }
              seen[i]++;
              rec = b.tris[i];
            }
            for (int k = 0; k < 3; ++k) {
              const float* q = sc.vertex_pos + 3 * (size_t)sc.tri_vtx[3 * (size_t)rec.id + k];
              for (int a = 0; a < 3; ++a) r.lo[a] = std::min(r.lo[a], q[a]), r.hi[a] = std::max(r.hi[a], q[a]);
            }
          }
          return r;
        }
        float lo[2][3], hi[2][3];
        int32_t child[2];
        if (q8) {
          if ((size_t)((uint32_t)ref >> 4) >= visited.size()) { err = "inner ref beyond the unified array"; return r; }
          if (visited[(uint32_t)ref >> 4]++) { err = "node reached twice"; return r; }
          rtbvh::decodeQ8(b, (uint32_t)ref, lo, hi, child);
        } else {
          if (ref % 32) { err = "inner ref is not a multiple of 32"; return r; }
          const uint32_t idx = (uint32_t)ref / 32u;
          if (idx >= visited.size()) { err = "inner ref beyond the node array"; return r; }
          if (visited[idx]++) { err = "node reached twice"; return r; }
          const rtbvh::Node16& n = b.nodes16[idx];
          for (int i = 0; i < 2; ++i) {
            const uint16_t* q = i ? n.box1 : n.box0;
            for (int a = 0; a < 3; ++a) lo[i][a] = rtbvh::halfToFloat(q[2 * a]) * inv, hi[i][a] = rtbvh::halfToFloat(q[2 * a + 1]) * inv;
            child[i] = n.child[i];
          }
        }
        ++nodes;
        for (int i = 0; i < 2 && err.empty(); ++i) {
          const Bx c = walk(child[i], depth + 1u);
          if (child[i] >= 0) {
            const double dx = (double)hi[i][0] - lo[i][0], dy = (double)hi[i][1] - lo[i][1], dz = (double)hi[i][2] - lo[i][2];
            areaSum += dx * dy + dy * dz + dz * dx;
          }
          for (int a = 0; a < 3; ++a) {
            // the stored box must contain the geometry below it, padded
            if (!(lo[i][a] <= c.lo[a] - 0.999f * b.pad && hi[i][a] >= c.hi[a] + 0.999f * b.pad)) err = "a child box does not contain its padded geometry";
            r.lo[a] = std::min(r.lo[a], c.lo[a]), r.hi[a] = std::max(r.hi[a], c.hi[a]);
          }
        }
        return r;
      }
    } W{*sc, b, seen, visited, depthSeen, areaSum, inv, q8, 0, {}};
    W.walk(q8 ? (int32_t)rtbvh::kQ8RootOffset : 0, 0);
    if (!W.err.empty()) return fail(RT_ERR_STATE, "BVH check: %s", W.err.c_str());
    // (a one-triangle scene has that triangle under both children of its root: bvh_build.cpp build())
    for (uint32_t v : seen)
      if (v != 1 && !(sc->n_triangles == 1 && v == 2)) return fail(RT_ERR_STATE, "BVH check: a triangle record is referenced %u times", v);
    if (W.nodes != nn) return fail(RT_ERR_STATE, "BVH check: %llu of %zu nodes reachable", (unsigned long long)W.nodes, nn);
    if (depthSeen != b.maxDepth) return fail(RT_ERR_STATE, "BVH check: deepest leaf at level %u, builder says %u", depthSeen, b.maxDepth);
    if ((int)depthSeen > b.depthCap) return fail(RT_ERR_STATE, "BVH check: depth %u exceeds the cap %d", depthSeen, b.depthCap);
    out8[0] = (uint32_t)nn, out8[1] = (uint32_t)b.q8.size(), out8[2] = b.q8Blocks, out8[3] = b.maxDepth;
    out8[4] = out8[5] = out8[6] = out8[7] = 0;
    // the surface-area estimate of node visits per random ray: the root plus every inner child by its box area
    if (est2) {
      auto area = [](const float* lo, const float* hi) {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
      };
      const rtbvh::Node& n0 = b.nodes[0];
      float lo[3], hi[3];
      for (int a = 0; a < 3; ++a) lo[a] = std::min(n0.lo0[a], n0.lo1[a]), hi[a] = std::max(n0.hi0[a], n0.hi1[a]);
      const double rootArea = std::max(area(lo, hi), 1e-300);
      double v = 0;
      for (const rtbvh::Node& n : b.nodes) {
        if (n.child[0] >= 0) v += area(n.lo0, n.hi0);
        if (n.child[1] >= 0) v += area(n.lo1, n.hi1);
      }
      est2[0] = 1.0 + v / rootArea, est2[1] = 1.0 + areaSum / rootArea;
    }
  } catch (const std::exception& e) {
    return fail(RT_ERR_INVALID, "BVH check failed: %s", e.what());
  }
  return RT_OK;
}

int rt_bvh_top_check_host(const rt_scene_desc* sc, uint32_t leaf_max, uint32_t cutoff, uint32_t* out8) {
  if (!sc || !out8) return fail(RT_ERR_INVALID, "null argument");
  try {
    rtbvh::TopBuilt t;
    rtbvh::buildTop(*sc, leaf_max, cutoff, t);
    const uint32_t n = sc->n_triangles;
    if (t.order.size() != n) return fail(RT_ERR_STATE, "top check: the order has %zu entries for %u triangles", t.order.size(), n);
    std::vector<uint8_t> seenTri(n, 0), covered(n, 0), partRef(t.parts.size(), 0);
    for (uint32_t id : t.order) {
      if (id >= n || seenTri[id]++) return fail(RT_ERR_STATE, "top check: the order is not a permutation");
    }
    auto geomBox = [&](uint32_t b, uint32_t e, float* lo, float* hi) {
      for (int a = 0; a < 3; ++a) lo[a] = 3e38f, hi[a] = -3e38f;
      for (uint32_t i = b; i < e; ++i)
        for (int k = 0; k < 3; ++k) {
          const float* q = sc->vertex_pos + 3 * (size_t)sc->tri_vtx[3 * (size_t)t.order[i] + k];
          for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], q[a]), hi[a] = std::max(hi[a], q[a]);
        }
    };
    uint32_t largest = 0, deepest = 0, topLeaves = 0;
    for (uint32_t i = 0; i < t.nodes.size(); ++i)
      for (int c = 0; c < 2; ++c) {
        const int32_t ref = t.nodes[i].child[c];
        const float* blo = c ? t.nodes[i].lo1 : t.nodes[i].lo0;
        const float* bhi = c ? t.nodes[i].hi1 : t.nodes[i].hi0;
        uint32_t b = 0, e = 0;
        if (ref >= 0) {
          if ((uint32_t)ref >= t.nodes.size() || (uint32_t)ref == i) return fail(RT_ERR_STATE, "top check: bad inner ref");
          continue;
        }
        const uint32_t code = ~(uint32_t)ref;
        if (code & rtbvh::kPartFlag) {
          const uint32_t k = code & (rtbvh::kPartFlag - 1u);
          if (k >= t.parts.size() || partRef[k]++) return fail(RT_ERR_STATE, "top check: part %u referred to twice or out of range", k);
          const rtbvh::TopBuilt::Part& P = t.parts[k];
          if (P.parent != i || P.slot != (uint32_t)c) return fail(RT_ERR_STATE, "top check: part %u names another referrer", k);
          b = P.b, e = P.e;
          if (e <= b || e > n || e - b > cutoff || e - b <= t.leafMax) return fail(RT_ERR_STATE, "top check: part %u has %u triangles", k, e - b);
          largest = std::max(largest, e - b), deepest = std::max(deepest, P.depth);
          // its subtree must still fit below: ceil(log2(triangles / leafMax)) more levels at least
          uint32_t need = 0;
          for (uint32_t m = e - b; m > t.leafMax; m = (m + 1) / 2) ++need;
          if ((int)(P.depth + need) > t.depthCap) return fail(RT_ERR_STATE, "top check: part %u at depth %u cannot be split within the cap %d", k, P.depth, t.depthCap);
        } else {
          b = code >> 3, e = b + (code & 7u) + 1u;
          if (e > n || e - b > t.leafMax) return fail(RT_ERR_STATE, "top check: bad leaf");
          ++topLeaves;
        }
        for (uint32_t j = b; j < e; ++j)
          if (covered[j]++) return fail(RT_ERR_STATE, "top check: position %u of the order is covered twice", j);
        float lo[3], hi[3];
        geomBox(b, e, lo, hi);
        for (int a = 0; a < 3; ++a)
          if (!(blo[a] <= lo[a] - 0.999f * t.pad && bhi[a] >= hi[a] + 0.999f * t.pad)) return fail(RT_ERR_STATE, "top check: a child box does not contain its padded geometry");
      }
    for (uint32_t j = 0; j < n; ++j)
      if (covered[j] != 1) return fail(RT_ERR_STATE, "top check: position %u of the order is not covered", j);
    for (size_t k = 0; k + 1 < t.parts.size(); ++k)
      if (t.parts[k].b >= t.parts[k + 1].b) return fail(RT_ERR_STATE, "top check: parts are not in order");
    out8[0] = (uint32_t)t.nodes.size(), out8[1] = (uint32_t)t.parts.size(), out8[2] = largest, out8[3] = deepest;
    out8[4] = (uint32_t)t.depthCap, out8[5] = topLeaves, out8[6] = out8[7] = 0;
  } catch (const std::exception& e) {
    return fail(RT_ERR_INVALID, "top check failed: %s", e.what());
  }
  return RT_OK;
}

int rt_bvh_tune(rt_ctx* c, const rt_params* probe, double budget_seconds, uint32_t max_probes, rt_tune_report* out) {
  if (!c || !probe) return fail(RT_ERR_INVALID, "ctx/probe is null");
  if (out) memset(out, 0, sizeof *out);
  if (c->builder != RT_BVH_HOST || c->bvh.nodes.empty() || c->nodeFormat != RT_NODES_F16)
    return fail(RT_ERR_STATE, "rt_bvh_tune needs a host-built tree in the RT_NODES_F16 format");
  if (probe->use_photons || probe->accel != RT_ACCEL_BVH) return fail(RT_ERR_INVALID, "the probe must be a BVH render without the photon map");
  int rc = check_params(c, probe);
  if (rc != RT_OK) return rc;
  if (!(budget_seconds > 0)) return RT_OK;
  HIP_TRY(hipSetDevice(c->device));
  float4* dAcc = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dAcc), (size_t)probe->width * probe->height * sizeof(float4)));
  rt_params p = *probe;
  p.collect_stats = 1;
  const size_t nodeBytes = c->bvh.nodes.size() * sizeof(rtbvh::Node16);
  std::string err;
  auto upload = [&]() -> bool {
    if (hipMemcpy(const_cast<uint4*>(c->S.nodes), c->bvh.nodes16.data(), nodeBytes, hipMemcpyHostToDevice) != hipSuccess) {
      err = "node upload failed";
      return false;
    }
    return true;
  };
  auto measure = [&]() -> double {
    if (!err.empty()) return 1e300;
    rtbvh::packNodes(c->bvh);
    if (!upload()) return 1e300;
    if (hipMemset(dAcc, 0, (size_t)p.width * p.height * sizeof(float4)) != hipSuccess) {
      err = "probe accumulator reset failed";
      return 1e300;
    }
    rt_stats st;
    if (rt_render_device(c, &p, dAcc, nullptr, &st) != RT_OK) {
      err = std::string("probe render failed: ") + rt_last_error();
      return 1e300;
    }
    return (double)st.nodes_visited + 1.5 * (double)st.tris_tested;
  };
  // A second probe (another seed: other jitter, other light samples, other bounce directions) is the referee: a tuned
  // tree that does not also beat the original on rays it was not tuned on is dropped (over-fitting shows on scenes
  // whose probe is too small for their triangle count).
  rtbvh::TuneReport rep;
  bool kept = true;
  const std::vector<rtbvh::Node> original = c->bvh.nodes;
  const uint32_t depth0 = c->bvh.maxDepth;
  try {
    rt_params pv = p;
    pv.seed = p.seed ^ 0x9e3779b9u;
    auto referee = [&]() {
      const rt_params keep = p;
      p = pv;
      const double v = measure();
      p = keep;
      return v;
    };
    const double v0 = referee();
    rep = rtbvh::tuneMeasured(c->bvh, measure, budget_seconds, max_probes, getenv("RT_BVH_VERBOSE") != nullptr);
    const double v1 = referee();
    if (getenv("RT_BVH_VERBOSE")) fprintf(stderr, "tune referee probe: %.6g -> %.6g\n", v0, v1);
    if (!(v1 < v0)) c->bvh.nodes = original, c->bvh.maxDepth = depth0, kept = false;
    // final numbering (the LDS-resident prefix is chosen by area from the root) and the device copy
    rtbvh::relayoutAndPack(c->bvh);
  } catch (const std::exception& e) {
    err = e.what();
  }
  if (!err.empty()) {
    // a failed probe, upload or builder step: the context goes back to the tree it came with, host AND device side
    // (the device may hold whichever candidate was uploaded last)
    const std::string why = err;
    err.clear();
    try {
      c->bvh.nodes = original, c->bvh.maxDepth = depth0;
      rtbvh::relayoutAndPack(c->bvh);
    } catch (const std::exception& e) {
      err = e.what();
    }
    if (err.empty()) upload();
    (void)hipFree(dAcc);
    if (!err.empty()) {  // not even that: the device tree is unknown, the context must not render again
      c->broken = true;
      return fail(RT_ERR_STATE, "rt_bvh_tune: %s; restoring the original tree failed too (%s): the context is unusable", why.c_str(), err.c_str());
    }
    return fail(RT_ERR_HIP, "rt_bvh_tune: %s (the original tree is back in place)", why.c_str());
  }
  upload();
  (void)hipFree(dAcc);
  if (!err.empty()) {
    c->broken = true;
    return fail(RT_ERR_STATE, "rt_bvh_tune: %s: the context is unusable", err.c_str());
  }
  if (out) out->probes = rep.probes, out->accepted = kept ? rep.accepted : 0u, out->cost_before = rep.cost0, out->cost_after = kept ? rep.cost1 : rep.cost0, out->seconds = rep.seconds;
  return RT_OK;
}

int rt_profile_reset(rt_ctx* c) {
  if (!c) return fail(RT_ERR_INVALID, "ctx is null");
  c->evUsed = 0;
  return RT_OK;
}

int rt_profile_collect(rt_ctx* c, double* total_ms, uint32_t* launches) {
  if (!c || !total_ms || !launches) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  const int n = c->evUsed < kEventPairs ? c->evUsed : kEventPairs;
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]));
    sum += ms;
  }
  *total_ms = sum, *launches = static_cast<uint32_t>(n);
  return RT_OK;
}

int rt_test_unit(int32_t device, uint32_t which, const void* in, void* out, uint32_t n) {
  static const uint32_t inBytes[] = {8, 4, 4, 16, 60, 68, 56, 96, 112, 88, 8, 4};
  static const uint32_t outBytes[] = {8, 4, 4, 4, 16, 12, 24, 12, 48, 16, 16, 16};
  if (which > RT_UNIT_RECIP) return fail(RT_ERR_INVALID, "unknown unit %u", which);
  if (n && (!in || !out)) return fail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  int rc = select_device(device);
  if (rc != RT_OK) return rc;
  void *dIn = nullptr, *dOut = nullptr;
  HIP_TRY(hipMalloc(&dIn, (size_t)n * inBytes[which]));
  hipError_t he = hipMalloc(&dOut, (size_t)n * outBytes[which]);
  if (he == hipSuccess) he = hipMemcpy(dIn, in, (size_t)n * inBytes[which], hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(dOut, out, (size_t)n * outBytes[which], hipMemcpyHostToDevice);
  if (he == hipSuccess) he = rtk::launch_unit(which, dIn, dOut, n, nullptr);
  if (he == hipSuccess) he = hipMemcpy(out, dOut, (size_t)n * outBytes[which], hipMemcpyDeviceToHost);
  (void)hipFree(dIn);
  if (dOut) (void)hipFree(dOut);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "unit kernel failed: %s", hipGetErrorString(he));
  return RT_OK;
}

// ------------------------------------------------------------------ ray streams
int rt_trace_stream_device(rt_ctx* c, const void* d_ray_o, const void* d_ray_d, uint32_t n, void* d_res, void* stream) {
  if (!c || (n && (!d_ray_o || !d_ray_d || !d_res))) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  const uint32_t levels = (c->bvh.maxDepth > 1 ? c->bvh.maxDepth : 1) + 1u;
  rtk::DevScene Su = c->S;
  Su.slowRecip = 1u;  // the caller's rays: any length
  hipError_t he = rtk::launch_trace_stream(Su, static_cast<const float4*>(d_ray_o), static_cast<const float4*>(d_ray_d), n,
                                           static_cast<uint2*>(d_res), c->dTileCounter, levels, c->numCUs,
                                           static_cast<hipStream_t>(stream));
  if (he != hipSuccess) return fail(RT_ERR_HIP, "stream trace launch failed: %s", hipGetErrorString(he));
  return RT_OK;
}

// ------------------------------------------------------------------ photon map on the device
int rt_build_photon_map(rt_ctx* c, uint32_t n_requested, uint32_t seed, uint32_t* n_stored, double* ms_out) {
  if (!c || !n_stored) return fail(RT_ERR_INVALID, "ctx/n_stored is null");
  *n_stored = 0;
  HIP_TRY(hipSetDevice(c->device));
  // forget the old map first (see rt_set_photons)
  c->S.phPos = c->S.phDir = nullptr, c->S.phTopo = nullptr, c->S.n_photons = 0;
  if (c->phPos) (void)hipFree(c->phPos);
  if (c->phDir) (void)hipFree(c->phDir);
  if (c->phTopo) (void)hipFree(c->phTopo);
  c->phPos = c->phDir = nullptr, c->phTopo = nullptr;
  if (n_requested == 0 || c->S.n_lights == 0) return RT_OK;
  const float lightPdf = 1.f / static_cast<float>(c->S.n_lights);  // PhotonMap.h:19-20
  const uint32_t perLight = static_cast<uint32_t>(static_cast<int>(static_cast<float>(static_cast<int>(n_requested)) * lightPdf));
  const uint32_t n = perLight * c->S.n_lights;
  if (n == 0) return RT_OK;
  float4 *slotPos = nullptr, *slotDir = nullptr, *items = nullptr, *phPos = nullptr, *phDir = nullptr;
  uint4* phTopo = nullptr;
  uint32_t* dCount = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  auto cleanup = [&]() {
    for (void* p : {(void*)slotPos, (void*)slotDir, (void*)items, (void*)dCount})
      if (p) (void)hipFree(p);
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  };
  hipError_t he = hipMalloc(reinterpret_cast<void**>(&slotPos), n * sizeof(float4));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&slotDir), n * sizeof(float4));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&items), n * sizeof(float4));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dCount), sizeof(uint32_t));
  for (auto& e : ev)
    if (he == hipSuccess) he = hipEventCreate(&e);
  if (he == hipSuccess) he = hipEventRecord(ev[0], nullptr);
  if (he == hipSuccess) he = rtk::launch_emit(c->S, perLight, seed, slotPos, slotDir, c->dCounters, nullptr);
  if (he == hipSuccess) he = rtk::launch_photon_compact(slotPos, n, items, dCount, nullptr);
  if (he == hipSuccess) he = hipEventRecord(ev[1], nullptr);
  uint32_t m = 0;
  if (he == hipSuccess) he = hipMemcpy(&m, dCount, sizeof m, hipMemcpyDeviceToHost);  // (the count only)
  if (he == hipSuccess && m) {
    he = rtk::launch_kd_build(items, m, -1, nullptr);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&phPos), m * sizeof(float4));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&phDir), m * sizeof(float4));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&phTopo), 2 * (size_t)m * sizeof(uint4));
    if (he == hipSuccess) he = rtk::launch_photon_gather(items, slotDir, m, phPos, phDir, nullptr, nullptr);
    if (he == hipSuccess) he = rtk::launch_kd_topology(phPos, m, phTopo, nullptr);
  }
  if (he == hipSuccess) he = hipEventRecord(ev[2], nullptr);
  if (he == hipSuccess) he = hipDeviceSynchronize();
  if (he == hipSuccess && ms_out) {
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    ms_out[0] = a, ms_out[1] = b;  // emission + compaction, kd order + gather
  }
  cleanup();
  if (he != hipSuccess) {
    if (phPos) (void)hipFree(phPos);
    if (phDir) (void)hipFree(phDir);
    if (phTopo) (void)hipFree(phTopo);
    return fail(RT_ERR_HIP, "photon map build failed: %s", hipGetErrorString(he));
  }
  c->phPos = phPos, c->phDir = phDir, c->phTopo = phTopo;
  c->S.phPos = phPos, c->S.phDir = phDir, c->S.phTopo = phTopo, c->S.n_photons = m;
  *n_stored = m;
  return RT_OK;
}

int rt_get_photons(rt_ctx* c, float* pos3, float* dir3, float* weight, uint32_t cap, uint32_t* n_out) {
  if (!c || !n_out) return fail(RT_ERR_INVALID, "ctx/n_out is null");
  const uint32_t n = c->S.n_photons;
  *n_out = n;
  if (n == 0) return RT_OK;
  if (cap < n) return fail(RT_ERR_INVALID, "capacity %u < %u photons", cap, n);
  HIP_TRY(hipSetDevice(c->device));
  std::vector<float4> p(n), d(n);
  HIP_TRY(hipMemcpy(p.data(), c->phPos, n * sizeof(float4), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(d.data(), c->phDir, n * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    if (pos3) pos3[3 * (size_t)i] = p[i].x, pos3[3 * (size_t)i + 1] = p[i].y, pos3[3 * (size_t)i + 2] = p[i].z;
    if (dir3) dir3[3 * (size_t)i] = d[i].x, dir3[3 * (size_t)i + 1] = d[i].y, dir3[3 * (size_t)i + 2] = d[i].z;
    if (weight) weight[i] = d[i].w;
  }
  return RT_OK;
}

int rt_test_kd_order(int32_t device, const float* pos3, uint32_t n, int32_t depth_limit, uint32_t* perm_out, double* ms_out) {
  if (n && (!pos3 || !perm_out)) return fail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  int rc = select_device(device);
  if (rc != RT_OK) return rc;
  std::vector<float4> h(n);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t bits = i;
    float f;
    memcpy(&f, &bits, 4);
    h[i] = make_float4(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2], f);
  }
  float4 *items = nullptr, *scratch = nullptr;
  uint32_t* dPerm = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  rc = upload(&items, h.data(), n);
  if (rc != RT_OK) return rc;
  hipError_t he = hipMalloc(reinterpret_cast<void**>(&scratch), n * sizeof(float4));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dPerm), n * sizeof(uint32_t));
  if (he == hipSuccess) he = hipEventCreate(&e0);
  if (he == hipSuccess) he = hipEventCreate(&e1);
  if (he == hipSuccess) he = hipEventRecord(e0, nullptr);
  if (he == hipSuccess) he = rtk::launch_kd_build(items, n, depth_limit, nullptr);
  if (he == hipSuccess) he = hipEventRecord(e1, nullptr);
  if (he == hipSuccess) he = rtk::launch_photon_gather(items, nullptr, n, scratch, nullptr, dPerm, nullptr);
  if (he == hipSuccess) he = hipMemcpy(perm_out, dPerm, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  if (he == hipSuccess && ms_out) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms;
  }
  (void)hipFree(items);
  if (scratch) (void)hipFree(scratch);
  if (dPerm) (void)hipFree(dPerm);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "kd order failed: %s", hipGetErrorString(he));
  return RT_OK;
}

// ------------------------------------------------------------------ multi-GPU frame assembly
int rt_owned_granules(const rt_params* p, uint32_t rank, uint32_t* n_out) {
  if (!p || !n_out) return fail(RT_ERR_INVALID, "null argument");
  if (p->tile % 8 != 0) return fail(RT_ERR_INVALID, "tile must be a multiple of 8");
  owned_granules(p->width, p->height, rank, p->world, p->tile, nullptr, n_out);
  return RT_OK;
}

int rt_pack_owned_device(rt_ctx* c, const rt_params* p, const void* d_accum, void* d_packed, void* stream) {
  if (!c || !p || !d_accum || !d_packed) return fail(RT_ERR_INVALID, "null argument");
  if (p->tile % 8 != 0 || (p->world > 1 && p->rank >= p->world)) return fail(RT_ERR_INVALID, "bad rank/world/tile");
  HIP_TRY(hipSetDevice(c->device));
  rt_ctx::GranList L;
  int rc = ensure_granules(c, p, p->rank, &L);
  if (rc != RT_OK) return rc;
  hipError_t he = rtk::launch_pack(false, static_cast<const float4*>(d_accum), static_cast<float4*>(d_packed), L.d, L.n, p->width,
                                   p->height, static_cast<hipStream_t>(stream));
  if (he != hipSuccess) return fail(RT_ERR_HIP, "pack launch failed: %s", hipGetErrorString(he));
  return RT_OK;
}

int rt_unpack_owned_device(rt_ctx* c, const rt_params* p, uint32_t from_rank, const void* d_packed, void* d_accum, void* stream) {
  if (!c || !p || !d_accum || !d_packed) return fail(RT_ERR_INVALID, "null argument");
  if (p->tile % 8 != 0 || from_rank >= (p->world ? p->world : 1)) return fail(RT_ERR_INVALID, "bad rank/world/tile");
  HIP_TRY(hipSetDevice(c->device));
  rt_ctx::GranList L;
  int rc = ensure_granules(c, p, from_rank, &L);
  if (rc != RT_OK) return rc;
  hipError_t he = rtk::launch_pack(true, static_cast<const float4*>(d_packed), static_cast<float4*>(d_accum), L.d, L.n, p->width,
                                   p->height, static_cast<hipStream_t>(stream));
  if (he != hipSuccess) return fail(RT_ERR_HIP, "unpack launch failed: %s", hipGetErrorString(he));
  return RT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ rt_group: N devices, one process
// RCCL is reached through dlopen: single-GPU users of librt_amd.so never load it.
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name))
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return CommInitAll && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString;
  }
};
Rccl g_rccl;
}  // namespace

struct rt_group {
  std::vector<rt_ctx*> ctx;
  std::vector<int> dev;
  std::vector<hipStream_t> stream;
  hipStream_t xstream = nullptr;  // device 0: receives and scatters the other ranks' granules beside rank 0's own render
  hipEvent_t assembled = nullptr; // ... and tells stream[0] when the frame is whole
  std::vector<hipEvent_t> packed_ready;
  std::vector<float4*> accum;   // [rank] full frame on that rank's device
  std::vector<float4*> packed;  // [rank] owned granules, on that rank's device (rank > 0)
  std::vector<float4*> recv;    // [rank] the same bytes on device 0 (rank > 0)
  std::vector<ncclComm_t> comm; // RCCL communicators (all devices distinct), else empty: peer copies
  float *dBg = nullptr, *dOut = nullptr;
  size_t npx = 0;
  uint32_t fw = 0, fh = 0, ftile = 0;  // what the frame buffers are sized for
  uint32_t tile = 32;
  bool rccl = false;
};

namespace {
void group_free_frame(rt_group* g) {
  for (size_t r = 0; r < g->ctx.size(); ++r) {
    (void)hipSetDevice(g->dev[r]);
    if (g->accum[r]) (void)hipFree(g->accum[r]);
    if (g->packed[r]) (void)hipFree(g->packed[r]);
    g->accum[r] = g->packed[r] = nullptr;
  }
  (void)hipSetDevice(g->dev[0]);
  for (auto& p : g->recv) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  if (g->dBg) (void)hipFree(g->dBg);
  if (g->dOut) (void)hipFree(g->dOut);
  g->dBg = g->dOut = nullptr;
  g->npx = 0, g->fw = g->fh = g->ftile = 0;
}
}  // namespace

extern "C" {

int rt_group_create(const rt_scene_desc* scene, const int32_t* devices, uint32_t n, const rt_options* opt, rt_group** out) {
  if (!scene || !devices || !out || n == 0) return fail(RT_ERR_INVALID, "scene/devices/out is null or n == 0");
  if (n > 64) return fail(RT_ERR_INVALID, "at most 64 devices");
  *out = nullptr;
  rt_group* g = new rt_group();
  bool distinct = true;
  for (uint32_t r = 0; r < n; ++r)
    for (uint32_t q = 0; q < r; ++q) distinct = distinct && devices[r] != devices[q];
  // (sized before the first context exists: the failure path below runs rt_group_destroy ->
  // group_free_frame, which walks these for every context created so far)
  g->accum.assign(n, nullptr), g->packed.assign(n, nullptr), g->recv.assign(n, nullptr);
  for (uint32_t r = 0; r < n; ++r) {
    rt_options o{};
    if (opt) o = *opt;
    o.device = devices[r];
    rt_ctx* c = nullptr;
    // the host tree is built by the first context and copied by the others
    const rtbvh::Built* shared = (r > 0 && g->ctx[0]->builder == RT_BVH_HOST) ? &g->ctx[0]->bvh : nullptr;
    int rc = create_ctx(scene, &o, shared, &c);
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    if (rc == RT_OK && (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess))
      rc = fail(RT_ERR_HIP, "stream/event creation failed on device %d", devices[r]);
    if (rc != RT_OK) {
      const std::string keep = g_err;
      // (the failing rank's own stream / event / context are not in the group yet)
      if (s) (void)hipStreamDestroy(s);
      if (e) (void)hipEventDestroy(e);
      if (c) rt_destroy(c);
      rt_group_destroy(g);
      g_err = keep;
      return rc;
    }
    g->ctx.push_back(c), g->dev.push_back(devices[r]), g->stream.push_back(s), g->packed_ready.push_back(e);
  }
  (void)hipSetDevice(g->dev[0]);
  if (hipStreamCreateWithFlags(&g->xstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&g->assembled, hipEventDisableTiming) != hipSuccess) {
    rt_group_destroy(g);
    return fail(RT_ERR_HIP, "exchange stream creation failed on device %d", devices[0]);
  }
  // exchange path: RCCL send/recv when every rank has its own device (ncclCommInitAll refuses
  // duplicates); ranks sharing a device (rehearsal on one GPU) use peer copies
  if (((n > 1 && distinct) || (n == 1 && getenv("RT_GROUP_FORCE_RCCL"))) && !getenv("RT_GROUP_NO_RCCL")) {
    if (!g_rccl.load()) {
      rt_group_destroy(g);
      return fail(RT_ERR_UNSUPPORTED, "librccl.so could not be loaded (%s); set RT_GROUP_NO_RCCL=1 for peer copies", dlerror());
    }
    g->comm.assign(n, nullptr);
    ncclResult_t nr = g_rccl.CommInitAll(g->comm.data(), static_cast<int>(n), g->dev.data());
    if (nr != ncclSuccess) {
      g->comm.clear();
      rt_group_destroy(g);
      return fail(RT_ERR_HIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(nr));
    }
    g->rccl = true;
  }
  *out = g;
  return RT_OK;
}

void rt_group_destroy(rt_group* g) {
  if (!g) return;
  if (!g->ctx.empty()) group_free_frame(g);
  for (ncclComm_t c : g->comm)
    if (c) (void)g_rccl.CommDestroy(c);
  if (!g->dev.empty()) (void)hipSetDevice(g->dev[0]);
  if (g->xstream) (void)hipStreamDestroy(g->xstream);
  if (g->assembled) (void)hipEventDestroy(g->assembled);
  for (size_t r = 0; r < g->ctx.size(); ++r) {
    (void)hipSetDevice(g->dev[r]);
    if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
    if (g->packed_ready[r]) (void)hipEventDestroy(g->packed_ready[r]);
    rt_destroy(g->ctx[r]);
  }
  delete g;
}

uint32_t rt_group_size(const rt_group* g) { return g ? static_cast<uint32_t>(g->ctx.size()) : 0; }
int rt_group_uses_rccl(const rt_group* g) { return g && g->rccl ? 1 : 0; }
rt_ctx* rt_group_ctx(rt_group* g, uint32_t rank) { return g && rank < g->ctx.size() ? g->ctx[rank] : nullptr; }

int rt_group_set_photons(rt_group* g, const float* pos3, const float* dir3, uint32_t n) {
  if (!g) return fail(RT_ERR_INVALID, "group is null");
  for (rt_ctx* c : g->ctx) {
    int rc = rt_set_photons(c, pos3, dir3, n);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_group_render(rt_group* g, const rt_params* p, const float* bg, float* out_rgb, float* accum_out, rt_stats* stats) {
  if (!g) return fail(RT_ERR_INVALID, "group is null");
  const uint32_t n = static_cast<uint32_t>(g->ctx.size());
  int rc = check_params(g->ctx[0], p);
  if (rc != RT_OK) return rc;
  if (out_rgb && !bg) return fail(RT_ERR_INVALID, "out_rgb requested without a background image");
  const size_t npx = (size_t)p->width * p->height;
  rt_params base = *p;
  base.world = n, base.tile = p->tile ? p->tile : g->tile;
  std::vector<uint32_t> cnt(n, 0);
  for (uint32_t r = 0; r < n; ++r) owned_granules(p->width, p->height, r, n, base.tile, nullptr, &cnt[r]);
  if (p->width != g->fw || p->height != g->fh || base.tile != g->ftile) {  // frame buffers for this image size and sharding
    group_free_frame(g);
    for (uint32_t r = 0; r < n; ++r) {
      HIP_TRY(hipSetDevice(g->dev[r]));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->accum[r]), npx * sizeof(float4)));
      if (r > 0 && cnt[r]) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->packed[r]), (size_t)cnt[r] * 64 * sizeof(float4)));
    }
    HIP_TRY(hipSetDevice(g->dev[0]));
    for (uint32_t r = 1; r < n; ++r)
      if (cnt[r]) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->recv[r]), (size_t)cnt[r] * 64 * sizeof(float4)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->dBg), npx * 3 * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->dOut), npx * 3 * sizeof(float)));
    g->npx = npx, g->fw = p->width, g->fh = p->height, g->ftile = base.tile;
  }
  // 1. every rank integrates its tiles (asynchronously, each on its own device and stream)
  std::vector<int> ev(n, 0);
  for (uint32_t r = 0; r < n; ++r) {
    rt_params pr = base;
    pr.rank = r;
    HIP_TRY(hipSetDevice(g->dev[r]));
    HIP_TRY(hipMemsetAsync(g->accum[r], 0, npx * sizeof(float4), g->stream[r]));
    if (r == 0) {  // the exchange stream scatters into this buffer: only after it has been zeroed
      HIP_TRY(hipEventRecord(g->assembled, g->stream[0]));
      HIP_TRY(hipStreamWaitEvent(g->xstream, g->assembled, 0));
    }
    HIP_TRY(hipMemsetAsync(g->ctx[r]->dCounters, 0, RTK_CNT_COUNT * sizeof(unsigned long long), g->stream[r]));
    rc = launch_frame(g->ctx[r], &pr, g->accum[r], g->stream[r], &ev[r]);
    if (rc != RT_OK) return rc;
    if (r > 0 && cnt[r]) {
      rc = rt_pack_owned_device(g->ctx[r], &pr, g->accum[r], g->packed[r], g->stream[r]);
      if (rc != RT_OK) return rc;
      HIP_TRY(hipEventRecord(g->packed_ready[r], g->stream[r]));
    }
  }
  // 2. owned granules travel to rank 0's device: 1/N of the frame per rank, nothing else.  Device 0
  // receives and scatters them on its EXCHANGE stream, beside its own render on stream[0] (the
  // pixels are disjoint), so a rank that finishes early is assembled while the others still render.
  if (g->rccl) {
    ncclResult_t nr = g_rccl.GroupStart();
    for (uint32_t r = 1; r < n && nr == ncclSuccess; ++r) {
      if (!cnt[r]) continue;
      const size_t floats = (size_t)cnt[r] * 64 * 4;
      nr = g_rccl.Send(g->packed[r], floats, ncclFloat, 0, g->comm[r], g->stream[r]);
      if (nr == ncclSuccess) nr = g_rccl.Recv(g->recv[r], floats, ncclFloat, static_cast<int>(r), g->comm[0], g->xstream);
    }
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return fail(RT_ERR_HIP, "RCCL exchange failed: %s", g_rccl.GetErrorString(nr));
  } else {
    HIP_TRY(hipSetDevice(g->dev[0]));
    for (uint32_t r = 1; r < n; ++r) {
      if (!cnt[r]) continue;
      HIP_TRY(hipStreamWaitEvent(g->xstream, g->packed_ready[r], 0));
      HIP_TRY(hipMemcpyPeerAsync(g->recv[r], g->dev[0], g->packed[r], g->dev[r], (size_t)cnt[r] * 64 * sizeof(float4), g->xstream));
    }
  }
  // 3. rank 0 scatters them into its frame, resolves (Renderer.cpp:262-265) and hands the image back
  HIP_TRY(hipSetDevice(g->dev[0]));
  for (uint32_t r = 1; r < n; ++r) {
    if (!cnt[r]) continue;
    rc = rt_unpack_owned_device(g->ctx[0], &base, r, g->recv[r], g->accum[0], g->xstream);
    if (rc != RT_OK) return rc;
  }
  HIP_TRY(hipEventRecord(g->assembled, g->xstream));
  HIP_TRY(hipStreamWaitEvent(g->stream[0], g->assembled, 0));
  hipError_t he = hipSuccess;
  if (out_rgb) {
    he = hipMemcpyAsync(g->dBg, bg, npx * 3 * sizeof(float), hipMemcpyHostToDevice, g->stream[0]);
    if (he == hipSuccess) rc = rt_resolve_device(g->ctx[0], p->width, p->height, p->spp, g->accum[0], g->dBg, g->dOut, g->stream[0]);
    if (rc != RT_OK) return rc;
    if (he == hipSuccess) he = hipMemcpyAsync(out_rgb, g->dOut, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, g->stream[0]);
  }
  if (he == hipSuccess && accum_out) he = hipMemcpyAsync(accum_out, g->accum[0], npx * sizeof(float4), hipMemcpyDeviceToHost, g->stream[0]);
  if (he != hipSuccess) return fail(RT_ERR_HIP, "frame read-back failed: %s", hipGetErrorString(he));
  for (uint32_t r = 0; r < n; ++r) {
    HIP_TRY(hipSetDevice(g->dev[r]));
    HIP_TRY(hipStreamSynchronize(g->stream[r]));
  }
  HIP_TRY(hipSetDevice(g->dev[0]));
  HIP_TRY(hipStreamSynchronize(g->xstream));
  if (stats) {
    memset(stats, 0, sizeof *stats);
    for (uint32_t r = 0; r < n; ++r) {
      HIP_TRY(hipSetDevice(g->dev[r]));
      rt_stats s{};
      rc = read_counters(g->ctx[r], &s);
      if (rc != RT_OK) return rc;
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, g->ctx[r]->ev[ev[r]][0], g->ctx[r]->ev[ev[r]][1]));
      stats->rays_closest += s.rays_closest, stats->rays_shadow += s.rays_shadow, stats->knn_queries += s.knn_queries;
      stats->nodes_visited += s.nodes_visited, stats->tris_tested += s.tris_tested, stats->kd_visited += s.kd_visited;
      stats->kernel_ms = ms > stats->kernel_ms ? ms : stats->kernel_ms;  // the slowest rank
    }
    stats->samples = (uint64_t)npx * (p->spp_count ? p->spp_count : p->spp);
  }
  return RT_OK;
}

}  // extern "C"
