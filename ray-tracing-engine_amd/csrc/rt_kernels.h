// rt_kernels.h — host<->device argument blocks and kernel launchers
// (implemented in rt_kernels.hip, called from rt_api.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_build.h"
#include "rt_amd.h"
#include "rt_device.h"

#define RTK_KMAX 16  // photon k-heap slots per lane (LDS budget, rt_kernels.hip)

enum {
  RTK_CNT_CLOSEST = 0,
  RTK_CNT_SHADOW,
  RTK_CNT_KNN,
  RTK_CNT_NODES,
  RTK_CNT_TRIS,
  RTK_CNT_KD,
  RTK_CNT_WNODE,  // wave-level node steps (a step = one descent iteration of a wave)
  RTK_CNT_WLEAF,  // wave-level leaf phases
  RTK_CNT_LWAIT,  // lanes holding a leaf, summed over the wave-level node steps
  RTK_CNT_LIDLE,  // lanes without a ray, summed over the wave-level node steps
  RTK_CNT_FRAMES, // Q8 node format: block frames fetched
  RTK_CNT_COUNT = 32  // [16..31]: section clocks of the diagnostic build (RT_PHASE_TIMING)
};

namespace rtk {

// Device-resident flattened scene (all pointers are HBM allocations of rt_ctx).
struct DevScene {
  const uint4* nodes;      // n_nodes x 32 B: 12 x f16 box planes ((lo, hi) per axis, child 0 then child 1, scaled) + 2 child refs
  const uint4* q8;         // the one-request form (rtbvh::Slot16: frames, 16-B node records and triangle records in one array), or null:
                           // when set, the pooled render kernel and k_trace traverse it instead of nodes / tris
  uint32_t q8ShiftBytes;   // log2 of a block's BYTES (rtbvh::Built::q8Shift + 4)
  const float4* tris;      // n_tris x 48 B, BVH leaf order: {p0,e1.x}{e1.yz,e2.xy}{e2.z,id,mesh,-}
  const float4* trisRef;   // same records in reference (mesh,tri) order (brute-force path)
  const uint4* triShade;   // per global triangle id: {v0,v1,v2 (global vertex ids), mesh}
  const float* vpos;       // [n_vertices][3]
  const float* vnrm;       // [n_vertices][3]
  const rt_material* mats; // [n_meshes]
  const rtd::DevMat* matsDev;  // [n_meshes] the same with the per-material constants of Material.h:25-70 worked out (rt_device.h)
  const rt_light* lights;  // [n_lights]
  const uint32_t* meshTriBegin;
  const uint32_t* meshVtxBegin;
  const float4* phPos;     // photons in kd-tree order: xyz + pad
  const float4* phDir;     // income direction xyz + weight
  const uint4* phTopo;     // per photon, 32 B: {position xyz, left child | axis << 30}{right child, parent's split coordinate, parent's axis, 0} (kd_build.hip k_kd_topology)
  uint32_t n_tris, n_nodes, n_lights, n_photons;
  float invBoxScale;       // 1 / rtbvh::Built::boxScale
  uint32_t topK;           // node records [0, topK) are LDS-resident in the persistent kernel (set per launch)
  float originBound;       // k_trace: rays starting farther out run the exhaustive loop (rtbvh::Built)
  uint32_t leafT;          // Trav::round leaves its descent when fewer lanes than this still descend ...
  uint32_t leafMul;        // ... and fewer than leafMul/64 of the wave's live lanes
  uint32_t slowRecip;      // 1: 1 / det by division (rays of unknown length); 0: rtd::recip_fast — |det| < 2^100 is guaranteed
  uint32_t refillT;        // vertex_pool hands out rays once this many workers are free
  uint32_t stealT;         // ... and splits the stacks of the last long rays once this many are free
  rt_camera cam;
};

struct RenderArgs {
  const uint32_t* tiles;  // owned wave tiles: x0 | y0 << 16 (pixels, top-left corner)
  uint32_t n_tiles;
  uint32_t width, height, spp, s0, s1, mode, max_depth, seed, k, photons_requested;
  uint32_t flags;         // bit 0: shadow rays through the wave-level pool
  uint32_t stackLevels;   // LDS traversal-stack rows (64 words each) per wave: BVH depth + 1; with photons also the kd walk's
                          // pending entries (kd depth + 2: 32-bit, or two 16-bit entries per word when kd16)
  uint32_t kd16;          // the k-NN walk keeps 16-bit stack entries (fewer than 65,535 photons)
  uint32_t sshift;        // a wave = (64 >> sshift) pixels x (1 << sshift) samples side by side
  uint32_t tileW, tileH;  // pixel footprint of one wave (tileW * tileH == 64 >> sshift)
  uint32_t tilesPerBlock; // one-wave-per-workgroup kernels (k_render): consecutive wave tiles a workgroup renders (launcher)
  // persistent pooled kernel (k_render_persist)
  uint32_t* tileCounter;  // next wave tile to hand out (zeroed before the launch)
  uint32_t waveWords;     // LDS words per wave (stack levels x 64 + pool), set by the launcher
  uint32_t numCUs;        // workgroups to launch (one per CU)
};

hipError_t launch_render(bool brute_force, bool photon, bool stats, const DevScene& S, const RenderArgs& A,
                         float4* accum, unsigned long long* counters, hipStream_t stream);
hipError_t launch_resolve(uint32_t n_pixels, uint32_t spp, const float4* accum, const float* bg, float* out,
                          hipStream_t stream);
// The wavefront integrator (wavefront.hip): path state of one batch, SoA in HBM
struct WfArgs {
  const uint32_t* gran;  // owned 8x8 granules (x8 | y8 << 16), row-major: path = (sample slot, granule, pixel)
  uint32_t nGran, width, height, spp, seed;
  uint32_t s0, s1;       // sample range still to do
  uint32_t batch;        // samples per batch
  uint32_t nPaths;       // paths of this batch (set by the launcher)
  uint32_t* rng;         // [P] engine state
  float4 *org, *dir;     // [P] current ray
  uint2* key;            // [P] current hit {t bits, triangle id}; {~0, ~0}: the path has ended
  float4 *nrm, *pnt;     // [P] vertex normal (+ mesh in w), point
  float4* col;           // [3][P] vertex colours; col[0].w: primary hit, col[1].w: slot holds a sample
  float4 *rayO, *rayD;   // [4P] ray queue
  uint2* res;            // [4P] results
  unsigned long long* stripes;  // [1024][2] striped ray counters (closest, shadow), zero between frames
};
hipError_t launch_wavefront(const DevScene& S, const WfArgs& W, uint32_t mode, uint32_t maxDepth, float4* accum,
                            unsigned long long* counters, uint32_t* queueCounter, uint32_t stackLevels, uint32_t numCUs,
                            hipStream_t stream);
// ray queue in HBM -> results (wavefront stage T)
hipError_t launch_trace_stream(const DevScene& S, const float4* rayO, const float4* rayD, uint32_t n, uint2* res,
                               uint32_t* counter, uint32_t stackLevels, uint32_t numCUs, hipStream_t stream);
// multi-GPU frame assembly: pack (unpack = false) a rank's owned granules out of a full-frame
// accumulator, or scatter a packed buffer back into one
hipError_t launch_pack(bool unpack, const float4* src, float4* dst, const uint32_t* gran, uint32_t n, uint32_t width,
                       uint32_t height, hipStream_t stream);
hipError_t launch_trace(bool brute_force, bool any, const DevScene& S, const rt_ray* rays, uint32_t n,
                        rt_hit* hits, unsigned long long* counters, hipStream_t stream);
hipError_t launch_knn(const DevScene& S, const float* q, uint32_t n, uint32_t k, uint32_t* idx, float* dist,
                      uint32_t* visited, hipStream_t stream);
hipError_t launch_emit(const DevScene& S, uint32_t perLight, uint32_t seed, float4* outPos, float4* outDir,
                       unsigned long long* counters, hipStream_t stream);
// scene BVH on the device (bvh_gpu.hip): arrays are hipMalloc'ed by the builder, owned by the caller
struct GpuBvh {
  uint4* nodes16 = nullptr;   // n_nodes x 32 B (what the kernels traverse)
  float4* nodesF = nullptr;   // n_nodes x 64 B (rtbvh::Node: inspection / export)
  float4* tris = nullptr;     // leaf order
  float4* trisRef = nullptr;  // reference order
  uint32_t n_nodes = 0, maxDepth = 0;
};
// the exact build: the host builder's split rules as kernels above the exact subtrees (hSizeKey: rtbvh::planSceneExact)
hipError_t gpu_bvh_build_exact(const float* dVpos, const uint4* dTriShade, const float* hSizeKey, uint32_t n_tris, const rtbvh::ScenePlan& plan,
                               GpuBvh* out, hipStream_t stream);
// the hybrid build: the host builder's top (rtbvh::buildTop), exact subtrees of its parts on the device
hipError_t gpu_bvh_build_over_top(const float* dVpos, const uint4* dTriShade, const float* hSizeKey, uint32_t n_tris, const rtbvh::TopBuilt& top,
                                  GpuBvh* out, hipStream_t stream);
// photon map on the device (kd_build.hip)
hipError_t launch_photon_compact(const float4* slots, uint32_t n, float4* items, uint32_t* count, hipStream_t stream);
hipError_t launch_kd_build(float4* items, uint32_t n, int depthOverride, hipStream_t stream);
// explicit child / parent-split records of the median-implicit tree over phPos (what knn_query walks)
hipError_t launch_kd_topology(const float4* phPos, uint32_t n, uint4* topo, hipStream_t stream);
hipError_t launch_photon_gather(const float4* items, const float4* slotDir, uint32_t n, float4* phPos, float4* phDir,
                                uint32_t* perm, hipStream_t stream);
hipError_t launch_unit(uint32_t which, const void* in, void* out, uint32_t n, hipStream_t stream);
// *dBad (zeroed by the caller) != 0 afterwards: rtd::recip_fast / sqrt_fast differ from the division / sqrtf on this device
hipError_t launch_selfcheck_recip(uint32_t* dBad, hipStream_t stream);

}  // namespace rtk
