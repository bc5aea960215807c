/* rt_amd.h — C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference (nikitakaraevv/ray-tracing-engine) has no plugin/FFI seam; the
 * natural boundary is `void Renderer::render(Image&)` (reference
 * source/Renderer.h:36, source/Renderer.cpp:203-272), called once from main
 * (source/Main.cpp:224).  Everything below is what a `Renderer::render` that
 * dispatches to the GPU binds to (see INTEGRATION.md for the host-side stub):
 *
 *   rt_create / rt_destroy      <- Renderer::Renderer(Scene&, ...) deep copy of the
 *                                  scene (Renderer.cpp:15-31): flattened snapshot
 *   rt_set_photons              <- PhotonMap + kdtree built in render()
 *                                  (Renderer.cpp:209-213; kdtree.h:60-69 order)
 *   rt_emit_photons             <- PhotonMap::PhotonMap (PhotonMap.h:14-50,92-155)
 *   rt_render                   <- the spp/y/x loop + resolve (Renderer.cpp:219-271)
 *   rt_render_device/_resolve_device : same, on caller-owned DEVICE buffers and a
 *                                  caller stream (multi-GPU tile sharding, bench)
 *   rt_trace                    <- RayTracer::rayTrace (RayTracer.h:27-53) test hook
 *   rt_knn                      <- kdtree::knearest (kdtree.h:180-195) test hook
 *
 * Conventions: plain C, int status (0 = RT_OK), caller-owned buffers, no C++
 * types or exceptions across the boundary, one host thread per context and ONE LAUNCH IN
 * FLIGHT per context: a context owns one work-queue head, one counter block and one
 * wavefront state block, so rt_render_device / rt_trace_stream_device calls on the same
 * context must be ordered on one stream (or synchronised) — use one context per stream.  There
 * is NO CPU fallback: every compute entry point fails with RT_ERR_NO_DEVICE if
 * no gfx950 device is usable.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2

enum {
  RT_OK = 0,
  RT_ERR_INVALID = 1,   /* bad argument / inconsistent scene            */
  RT_ERR_NO_DEVICE = 2, /* no usable HIP device (never falls back to CPU) */
  RT_ERR_HIP = 3,       /* a HIP runtime call failed                      */
  RT_ERR_UNSUPPORTED = 4,
  RT_ERR_STATE = 5      /* e.g. photon shading requested without photons  */
};

/* Material.h:62-64 (m_kd, m_alpha, m_albedo, m_F0) */
typedef struct rt_material {
  float kd, alpha;
  float albedo[3];
  float f0[3];
} rt_material;

/* LightSource.h:61-65; basis from the ctor (:19-33) computed by the host */
typedef struct rt_light {
  float position[3], color[3], vertical[3], horizontal[3], normal[3];
  float intensity, side, factor, ac, al, aq;
} rt_light;

/* Camera.h:34-40 (m_position, m_lowerLeftCorner, m_horizontal, m_vertical) */
typedef struct rt_camera {
  float position[3], lower_left[3], horizontal[3], vertical[3];
} rt_camera;

/* Flattened Scene (Scene.h:27-31, Mesh.h:136-141).  Triangles are listed in the
 * reference's (mesh, triangle) iteration order (RayTracer.h:32-35): that order
 * is the tie-break for equal hit distances. */
typedef struct rt_scene_desc {
  uint32_t n_meshes, n_vertices, n_triangles, n_lights;
  const float* vertex_pos;        /* [n_vertices][3], meshes concatenated           */
  const float* vertex_nrm;        /* [n_vertices][3]                                */
  const uint32_t* tri_vtx;        /* [n_triangles][3] GLOBAL vertex ids             */
  const uint32_t* mesh_tri_begin; /* [n_meshes+1] first triangle of each mesh       */
  const uint32_t* mesh_vtx_begin; /* [n_meshes+1] first vertex of each mesh         */
  const rt_material* materials;   /* [n_meshes]                                     */
  const rt_light* lights;         /* [n_lights]                                     */
  rt_camera camera;
} rt_scene_desc;

enum { RT_MODE_RAY = 0, RT_MODE_PATH = 1 };   /* -m, Renderer.h:9-10            */
enum { RT_RNG_LEGACY = 0, RT_RNG_PIXEL = 1 }; /* legacy = global serial engine:
                                                 CPU oracle only, GPU refuses  */
enum { RT_ACCEL_BVH = 0, RT_ACCEL_BRUTE = 1 };
enum { RT_TRACE_CLOSEST = 0, RT_TRACE_ANY = 1 };

typedef struct rt_options {
  int32_t device;          /* HIP device ordinal                              */
  uint32_t bvh_leaf_max;   /* 0 = default (2), max 8                          */
  uint32_t bvh_builder;    /* RT_BVH_*; AUTO = the device builder for scenes of 8,192 triangles and
                              more (the same tree as the host builder's, 2-14 x sooner), the host
                              builder below (1-3 ms either way).  The environment variable
                              RT_BVH_GPU=1|2|3 forces DEVICE | HYBRID | HOST                        */
  uint32_t node_format;    /* RT_NODES_*: the node records the pooled render kernel and rt_trace
                              traverse.  AUTO picks per scene; the hits are the same either way  */
  uint32_t reserved[4];
} rt_options;
/* RT_BVH_HOST:   SAH on up to 16 host threads (binned above 4,096 triangles, exact sweeps below, size axis, rotations).
 * RT_BVH_DEVICE: the same split rules as kernels down to parts of <= 1,024 triangles, each part one exact SAH subtree (one
 *                workgroup), then the host's rotation passes and a pre-order numbering as kernels: the SAME tree as
 *                RT_BVH_HOST's (equal boxes and leaf sets on the four preset scenes: tests/treedigest.py), 1 M triangles
 *                in about 20 ms.
 * RT_BVH_HYBRID: the host builder's own top, stopped at the same parts; everything below as RT_BVH_DEVICE.                 */
enum { RT_BVH_AUTO = 0, RT_BVH_DEVICE = 1, RT_BVH_HYBRID = 2, RT_BVH_HOST = 3 };
/* RT_NODES_F16: 32-byte records, 12 binary16 box planes + 2 child refs (two 16-byte requests per visit).
 * RT_NODES_Q8:  16-byte records, 12 8-bit box planes in the frame of the record's 16-KiB block + one
 *               packed child word (ONE request per visit), nodes and triangle records in one array.   */
enum { RT_NODES_AUTO = 0, RT_NODES_F16 = 1, RT_NODES_Q8 = 2 };

typedef struct rt_params {
  uint32_t width, height;
  uint32_t spp;        /* -N: samples per pixel of the whole frame             */
  uint32_t mode;       /* RT_MODE_*                                            */
  uint32_t max_depth;  /* 3 in the reference (Renderer.cpp:245,248)            */
  uint32_t seed;       /* stream key, include/rt_pixelmode.h                   */
  uint32_t rng_mode;   /* RT_RNG_PIXEL                                         */
  uint32_t accel;      /* RT_ACCEL_*                                           */
  uint32_t use_photons;       /* 1: photon-map shading (Renderer.cpp:63-104)   */
  uint32_t k;                 /* -k                                            */
  uint32_t photons_requested; /* -p: density denominator (Renderer.cpp:99)     */
  uint32_t spp_begin, spp_count; /* sample sub-range of this call; 0,0 = all   */
  uint32_t rank, world;       /* tile ownership: this call renders the 8x8-pixel
                                 tiles t with owner(t) == rank                 */
  uint32_t tile;              /* ownership granule in pixels (multiple of 8)   */
  uint32_t collect_stats;     /* 1: also count BVH nodes / triangle tests      */
  uint32_t reserved[7];       /* [0]: samples of a pixel one wave integrates side by
                                 side (power of two <= 64; 0 = chosen from the grid
                                 size).  [1] bit 0: shade vertices sequentially instead
                                 of through the wave's ray pool.  [2] bit 0: the queue-
                                 based (wavefront) integrator: path state and ray queues
                                 in HBM, one trace launch per path depth.  None of them
                                 changes the result, only the schedule.               */
} rt_params;

typedef struct rt_stats {
  uint64_t samples;        /* pixel-samples integrated                          */
  uint64_t rays_closest;   /* closest-hit casts (primary + bounce)              */
  uint64_t rays_shadow;    /* any-hit casts                                     */
  uint64_t knn_queries;
  uint64_t nodes_visited;  /* BVH node records fetched (collect_stats)          */
  uint64_t tris_tested;    /* 48-B triangle records tested (collect_stats)      */
  uint64_t kd_visited;     /* kd-tree nodes visited (collect_stats)             */
  uint64_t frame_fetches;  /* RT_NODES_Q8: block frames fetched (collect_stats) */
  double kernel_ms;        /* device time of the integrate kernel(s)            */
  uint64_t reserved[4];
} rt_stats;

typedef struct rt_ray {
  float origin[3], direction[3];
} rt_ray;

/* What RayTracer::rayTrace hands back (RayTracer.h:27-53) + the triangle id. */
typedef struct rt_hit {
  int32_t hit;         /* 0/1                                                   */
  uint32_t mesh;       /* meshIndex                                             */
  uint32_t tri;        /* triangle index inside the mesh                        */
  uint32_t vtx[3];     /* `triangle`: vertex ids LOCAL to the mesh              */
  float u, v, d;
} rt_hit;

typedef struct rt_bvh_info {
  uint32_t n_nodes, n_tri_records, max_depth, leaf_max;
  float pad;           /* absolute box padding used                             */
  float build_ms;      /* wall time of the build inside rt_create                */
  uint32_t builder;    /* RT_BVH_HOST / RT_BVH_DEVICE / RT_BVH_HYBRID            */
  uint32_t node_format;  /* RT_NODES_F16 / RT_NODES_Q8: what the pooled render kernel traverses */
  uint32_t flags;        /* RT_BVH_FLAG_*                                                         */
} rt_bvh_info;

/* SHORT_RECIP: the render instances take 1 / det and 1 / length in the three- / five-instruction forms that are
 * bit-identical to the IEEE operations inside the operand bounds rt_create checked (else they divide);
 * RECIP_CHECK_FAILED: this device's own check of those forms (2^25 inputs at its first rt_create) disagreed with
 * its division / sqrtf, so its contexts divide.  Same images either way.                                         */
enum { RT_BVH_FLAG_SHORT_RECIP = 1, RT_BVH_FLAG_RECIP_CHECK_FAILED = 2 };

typedef struct rt_ctx rt_ctx;

int rt_abi_version(void);
const char* rt_last_error(void);

/* Uploads the scene and builds its BVH.  RT_ERR_INVALID: null arrays / empty scene;
 * RT_ERR_UNSUPPORTED: 2^25 triangles or more (node and leaf refs are 31-bit byte offsets). */
int rt_create(const rt_scene_desc* scene, const rt_options* opt, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);

/* Photon arrays already in the host-built kd-tree (median-implicit) order. */
int rt_set_photons(rt_ctx* ctx, const float* pos3, const float* dir3, uint32_t n);
/* Emit photons on the GPU (pixel RNG mode, one stream per emitted photon), in
 * emission order; out arrays sized n_requested.  *n_out = stored photons. */
int rt_emit_photons(rt_ctx* ctx, uint32_t n_requested, uint32_t seed, float* pos3,
                    float* dir3, float* weight, uint32_t* n_out);

/* The whole photon map on the device (Renderer.cpp:209-213: PhotonMap + kdtree built inside
 * render()): emission (the rt_emit_photons kernel), stable compaction of the stored
 * particles, and the kd-tree order — the reference's recursive std::nth_element
 * (kdtree.h:60-69) restated for the GPU so that the array order, ties included, is the
 * library's (csrc/kd_build.hip) — installed as the context's photon map.  Photons never
 * visit the host; ms_out (optional, [2]) = {emission + compaction, kd order} device ms. */
int rt_build_photon_map(rt_ctx* ctx, uint32_t n_requested, uint32_t seed, uint32_t* n_stored,
                        double* ms_out);
/* The context's photon map in tree order (inspection / PhotonMap::saveToPCD). */
int rt_get_photons(rt_ctx* ctx, float* pos3, float* dir3, float* weight, uint32_t cap,
                   uint32_t* n_out);
/* Test hook: the device kd order of n host-given positions; perm_out[i] = input index of
 * tree slot i.  depth_limit < 0: std::nth_element's 2*lg(n); >= 0 forces the heap-select
 * path early (parity of __heap_select itself). */
int rt_test_kd_order(int32_t device, const float* pos3, uint32_t n, int32_t depth_limit,
                     uint32_t* perm_out, double* ms_out);

/* Whole frame on host buffers: background_rgb / out_rgb are [h][w][3] floats.
 * accum_out (optional, [h][w][4]) receives {sum r,g,b, primary-hit count}. */
int rt_render(rt_ctx* ctx, const rt_params* p, const float* background_rgb,
              float* out_rgb, float* accum_out, rt_stats* stats);

/* Progressive form of rt_render (Renderer.cpp:261-269: an image after every pass):
 * integrates the sample range p->spp_begin/spp_count on top of the caller-held host
 * accumulator accum_io ([h][w][4], zero before the first range) and resolves the
 * running estimate of the first spp_begin+spp_count samples into out_rgb. */
int rt_render_passes(rt_ctx* ctx, const rt_params* p, const float* background_rgb, float* accum_io,
                     float* out_rgb, rt_stats* stats);

/* Accumulate this rank's tiles / sample range into d_accum ([h][w][4] floats in
 * DEVICE memory, caller-zeroed) on `stream` (a hipStream_t, may be NULL). */
int rt_render_device(rt_ctx* ctx, const rt_params* p, void* d_accum, void* stream,
                     rt_stats* stats);
/* Renderer.cpp:262-265 on device buffers: out = sum/N + bg*(N-count)/N */
int rt_resolve_device(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp,
                      const void* d_accum, const void* d_background_rgb,
                      void* d_out_rgb, void* stream);

/* ---- multi-GPU (Renderer.cpp:219-265 sharded by pixel tiles) -------------
 * A tile-sharded frame: rank r of `world` integrates the pixels whose `tile`-pixel granule
 * (tx + ty) mod world == r (rt_params.rank/world/tile) into its own zeroed full-frame
 * accumulator.  Assembly moves only what a rank owns: its 8x8-pixel granules, packed
 * [granule][64] float4 in row-major granule order (n = rt_owned_granules), travel to the
 * assembling rank, which scatters them into its frame.  One process per GPU
 * (torch.distributed / MPI) uses the three calls below around its own collective;
 * one process driving N GPUs uses rt_group_*. */
int rt_owned_granules(const rt_params* p, uint32_t rank, uint32_t* n_out);
int rt_pack_owned_device(rt_ctx* ctx, const rt_params* p, const void* d_accum,
                         void* d_packed /* n*64 float4, device */, void* stream);
int rt_unpack_owned_device(rt_ctx* ctx, const rt_params* p, uint32_t from_rank,
                           const void* d_packed, void* d_accum, void* stream);

/* N devices driven from ONE host thread: the scene is replicated (rt_create per device),
 * every device integrates its tiles concurrently, owned granules go to devices[0] over
 * xGMI — RCCL ncclSend/ncclRecv (librccl.so, loaded on first use) when all devices are
 * distinct, peer copies when ranks share a device — and devices[0] resolves.  The image is
 * bit-identical to rt_render's on one device.  rt_params.rank/world are ignored (set per
 * device); tile = 0 selects 32. */
typedef struct rt_group rt_group;
int rt_group_create(const rt_scene_desc* scene, const int32_t* devices, uint32_t n,
                    const rt_options* opt, rt_group** out);
void rt_group_destroy(rt_group* g);
uint32_t rt_group_size(const rt_group* g);
int rt_group_uses_rccl(const rt_group* g);
rt_ctx* rt_group_ctx(rt_group* g, uint32_t rank); /* borrowed: e.g. rt_emit_photons on rank 0 */
int rt_group_set_photons(rt_group* g, const float* pos3, const float* dir3, uint32_t n);
int rt_group_render(rt_group* g, const rt_params* p, const float* background_rgb,
                    float* out_rgb, float* accum_out, rt_stats* stats);

/* Ray origins: the BVH's exactness argument (box padding vs the float triangle test's
 * error) covers origins up to 16 x max(|scene coordinate|, |camera|, |light position|);
 * rays that start farther out are answered by the exhaustive loop, transparently. */
int rt_trace(rt_ctx* ctx, const rt_ray* rays, uint32_t n, uint32_t accel,
             uint32_t kind, rt_hit* hits);
/* RayTracer::rayTrace over a ray QUEUE resident in HBM (the trace stage of the wavefront
 * integrator; also a device-to-device form of rt_trace): ray_o[i] = origin xyz + kind bits
 * in w (bit 0: any-hit), ray_d[i] = direction xyz (float4 each); res[i] (uint2) = closest
 * hit {t bits, global triangle id} or {~0, ~0}, any-hit {0 / 1, 0}.
 * Unlike rt_trace this form has NO exhaustive-loop fallback for far origins: every ray goes
 * through the BVH, so origins must lie within the range the padding covers (16 x max(|scene
 * coordinate|, |camera|, |light position|)) — true of every ray an integrator generates
 * (camera, surface points); the host cannot check device-resident rays. */
int rt_trace_stream_device(rt_ctx* ctx, const void* d_ray_o, const void* d_ray_d, uint32_t n,
                           void* d_res, void* stream);
int rt_knn(rt_ctx* ctx, const float* query3, uint32_t n, uint32_t k,
           uint32_t* idx_out /*[n][k]*/, float* dist_out /*[n][k]*/,
           uint32_t* visited_out /*[n] or NULL*/);

/* Inspection hooks for tests (host copies of the flattened acceleration data). */
int rt_bvh_info_get(rt_ctx* ctx, rt_bvh_info* out);
/* nodes64: the float (64-B) form of the node records — the device traverses the same
 * nodes packed to 32 B (binary16 planes rounded outward). */
int rt_bvh_export(rt_ctx* ctx, void* nodes64 /*n_nodes*64 B*/, void* tris48 /*n_tri_records*48 B*/);
/* The host BVH build alone (no GPU, no context): shape, FNV-1a digest of the node and
 * triangle arrays, and wall seconds.  threads: 0 = one per hardware thread (<= 16); the
 * digest must not depend on it.  (No reference counterpart: BVH.h:100-161 is dead code.) */
int rt_bvh_build_host(const rt_scene_desc* scene, uint32_t leaf_max, uint32_t threads, rt_bvh_info* info,
                      uint64_t* digest, double* seconds);
/* The host-built tree in the given device node format alone (no GPU): builds, packs and CHECKS the
 * result — every node reachable exactly once, every triangle record exactly once, every stored child box
 * containing its padded geometry, the depth as reported and within the cap.  node_format: RT_NODES_F16 or
 * RT_NODES_Q8.  out8 = {nodes, Q8: 16-byte slots of the unified array, Q8: blocks, depth, Q8: nodes the
 * packer had to add (leaf wrappers), 0, 0, 0}; est2 (optional) = surface-area estimate of node visits per
 * random ray from the float boxes / from the decoded boxes of the format.  (No reference counterpart: BVH.h
 * is dead code there.) */
int rt_bvh_check_host(const rt_scene_desc* scene, uint32_t leaf_max, uint32_t node_format, uint32_t* out8,
                      double* est2);
/* The host half of the hybrid builder alone (no GPU): the host builder's top down to parts of <= `cutoff` triangles,
 * CHECKED — the parts and the top's own leaves cover every triangle exactly once, every part is referred to by exactly one
 * child slot whose box contains its padded geometry, part roots lie within the depth cap with room for their subtrees, the
 * order is a permutation.  out8 = {top nodes, parts, largest part, deepest part root, depth cap, top leaves, 0, 0}.        */
int rt_bvh_top_check_host(const rt_scene_desc* scene, uint32_t leaf_max, uint32_t cutoff, uint32_t* out8);
/* Measured-cost tuning of the host-built BVH (no counterpart in the reference, whose rayTrace is the exhaustive
 * loop of RayTracer.h:27-53; this only changes HOW FAST the same hits are found).  Every tree over the same leaves
 * returns the same hits, so a probe frame traces exactly the same rays whatever the tree and its counters are a
 * deterministic cost of the tree for the rays of THIS scene, camera and integrator.  rt_bvh_tune renders `probe`
 * (a small frame: e.g. 128x128, 1-2 spp, the mode of the real render) once per proposed change — a subtree moved to
 * another place in the tree, the two children of a node in the other slot order — and keeps a change only if
 * nodes_visited + 1.5 * tris_tested fell.  Stops after `budget_seconds`, after `max_probes` probe frames (0 = no limit;
 * the counters are deterministic, so a probe limit — unlike a time limit — gives the same tree on every run) or when a
 * whole pass finds nothing; a second
 * probe with another seed referees the result (a tuned tree that is not better on it is dropped: accepted = 0).  Images
 * are unchanged by construction (and tested).  Needs a host-built binary tree (RT_ERR_STATE otherwise); must not run
 * concurrently with a launch on the same context. */
typedef struct rt_tune_report {
  uint32_t probes, accepted;
  double cost_before, cost_after; /* nodes_visited + 1.5 * tris_tested of the probe frame */
  double seconds;
  uint64_t reserved[4];
} rt_tune_report;
int rt_bvh_tune(rt_ctx* ctx, const rt_params* probe, double budget_seconds, uint32_t max_probes, rt_tune_report* out);
/* Device-time bookkeeping: every rt_render_device launch is bracketed by a HIP
 * event pair on its stream.  reset() forgets them; collect() synchronises the
 * device and returns the summed kernel time of the launches since reset
 * (at most 256 are remembered). */
int rt_profile_reset(rt_ctx* ctx);
int rt_profile_collect(rt_ctx* ctx, double* total_kernel_ms, uint32_t* launches);

/* Unit evaluations of single device building blocks on device `device` (parity
 * hooks for the reference's per-function golden vectors; also what the host
 * mirror's Ray::triangleIntersect / Material::evaluateColorResponse forward to).
 * Layouts (per element, 32-bit words unless noted):
 *   RT_UNIT_ASIN         in: double x                     out: double rt_asin(x)
 *   RT_UNIT_SINF/COSF    in: float x                      out: float
 *   RT_UNIT_STREAM_SEED  in: seed,domain,index,sub        out: uint32 state
 *   RT_UNIT_TRIANGLE     in: p0 p1 p2 origin dir (15 f)   out: hit(0/1 as float) u v t
 *                        (out is read first: u,v,t keep their input value where
 *                         Ray.cpp:9-24 leaves them unwritten)
 *   RT_UNIT_BSDF         in: kd alpha albedo3 f03 n3 wi3 wo3 (17 f)  out: rgb
 *   RT_UNIT_RAY_AT       in: camera(12 f) u v             out: origin3 dir3
 *   RT_UNIT_LIGHT_EVAL   in: rt_light(21 f) point3        out: rgb
 *   RT_UNIT_SAMPLERS     in: state idx N pad normal3 rt_light(21 f)  (28 words)
 *                        out: jitter xy, hemisphere dir3, light sample3, end state, pad3 (12 words)
 *   RT_UNIT_LIGHT_SAMPLE in: state rt_light(21 f) (22 words)  out: sample3, end state
 *                        (LightSource.h:46-49 randAreaPosition from a given engine state)
 *   RT_UNIT_POW          in: double x                     out: double x^2, x^5 (Material.h:38,48 pow)
 *   RT_UNIT_RECIP        in: float x                      out: the three-instruction reciprocal, 1.0f / x, the five-
 *                        instruction square root, sqrtf(x) (Ray.cpp:14's inv_det, Vec3.h:170-178's length: each pair
 *                        must agree bit for bit for 2^-100 <= |x| <= 2^100)
 */
enum {
  RT_UNIT_ASIN = 0,
  RT_UNIT_SINF = 1,
  RT_UNIT_COSF = 2,
  RT_UNIT_STREAM_SEED = 3,
  RT_UNIT_TRIANGLE = 4,
  RT_UNIT_BSDF = 5,
  RT_UNIT_RAY_AT = 6,
  RT_UNIT_LIGHT_EVAL = 7,
  RT_UNIT_SAMPLERS = 8,
  RT_UNIT_LIGHT_SAMPLE = 9,
  RT_UNIT_POW = 10,
  RT_UNIT_RECIP = 11
};
int rt_test_unit(int32_t device, uint32_t which, const void* in, void* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
