// wavefront_kernels.h — the queue-based form of the integrator.
//
// The same per-sample arithmetic as k_render (Renderer.cpp:227-258 + calculateColorRay /
// calculateColorPath :106-201), cut at the points where rays leave a vertex: path state
// lives in HBM, one lane owns one path in every kernel, and ALL secondary rays of a batch
// of paths go through one ray queue per path depth, traced by k_trace_stream (persistent
// waves refilled from the queue).  Per batch of B samples of every owned pixel:
//
//   wf_generate            RNG stream, jitter, camera ray (RayTracer.h:109-117, Camera.h:27-30),
//                          primary cast (coherent: traced in place)
//   for depth = 0..nvert-1
//     wf_vertex            hit -> normal, point (Renderer.cpp:42-43); light samples in light
//                          order, then the hemisphere sample (:52, :164): 3 shadow rays + the
//                          bounce ray into the queue [path][4]
//     k_trace_stream       RayTracer.h:27-53 for the whole queue
//     wf_finish            unoccluded lights -> BSDF x radiance (:56-59); the path moves on to
//                          the bounce hit or ends
//   wf_accumulate          clamp (:279-283) and add the batch's samples IN SAMPLE ORDER (:258)
//
// Every device function is the one k_render calls, draw order per stream is the same, so
// the frame is bit-identical to the fused kernel's and the oracle's (tests). It is NOT the
// default: on the bench scenes the trace stage alone is no faster than the whole fused frame
// (DESIGN.md §8); rt_params.reserved[2] = 1 selects it.
// (included by rt_kernels.hip inside namespace rtk: shares its device functions)
namespace {

constexpr uint32_t DEAD = 0xffffffffu;

__device__ __forceinline__ uint32_t wsum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// path i of a batch = (sample slot sb, owned granule g, pixel l of the granule)
__device__ __forceinline__ bool path_pixel(const WfArgs& W, uint32_t i, uint32_t& pix, uint32_t& px, uint32_t& py, uint32_t& smp) {
  const uint32_t perSample = W.nGran * 64u;
  const uint32_t sb = i / perSample, r = i - sb * perSample;
  const uint32_t g = r >> 6, l = r & 63u;
  const uint32_t gr = W.gran[g];
  px = (gr & 0xffffu) * 8u + (l & 7u), py = (gr >> 16) * 8u + (l >> 3);
  pix = py * W.width + px;
  smp = W.s0 + sb;
  return px < W.width && py < W.height && smp < W.s1;
}

}  // namespace

__global__ __launch_bounds__(64) void wf_generate(DevScene S, WfArgs W, unsigned long long* __restrict__ counters) {
  __shared__ uint32_t lds[(rtbvh::kMaxDepth + 1) * 64];
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  uint32_t pix = 0, px = 0, py = 0, smp = 0;
  const bool on = i < W.nPaths && path_pixel(W, i, pix, px, py, smp);
  Rng g{rt_stream_seed(W.seed, RT_STREAM_PIXEL, pix, smp)};
  float sx, sy;
  jitter_sample(g, (int)smp, (int)W.spp, sx, sy);
  f3 o, d;
  camera_ray(S.cam, ((float)px + sx) / (float)W.width, 1.f - ((float)py + sy) / (float)W.height, o, d);
  HitRec h;
  LaneStats st;
  const bool hit = traverse<false, false, LT_NONE>(S, on, o, d, lds + threadIdx.x, h, st);
  const float t = h.t;
  const uint32_t id = h.id;
  if (i < W.nPaths) {
    W.rng[i] = g.s;
    W.org[i] = make_float4(o.x, o.y, o.z, 0.f);
    W.dir[i] = make_float4(d.x, d.y, d.z, 0.f);
    W.key[i] = on && hit ? make_uint2(__float_as_uint(t), id) : make_uint2(DEAD, DEAD);
    // colours of the vertices never reached stay 0; w of col0 = "the primary ray hit" (Renderer.cpp:155-156)
    W.col[i] = make_float4(0.f, 0.f, 0.f, on && hit ? 1.f : 0.f);
    W.col[W.nPaths + i] = make_float4(0.f, 0.f, 0.f, on ? 1.f : 0.f);  // w: this slot is a real sample
    W.col[2 * (size_t)W.nPaths + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ray counts go to 1,024 striped slots (65 k waves adding to ONE address cost more than the
  // kernel itself: same-address atomics serialise at the memory side); wf_sum_stripes folds them
  const uint32_t c = wsum(on ? 1u : 0u);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&W.stripes[2u * (blockIdx.x & 1023u)], (unsigned long long)c);
}

// vertex of path depth `depth`: geometry, the stream's draws, the four rays
__global__ void wf_vertex(DevScene S, WfArgs W, uint32_t bounce, unsigned long long* __restrict__ counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = false;
  if (i < W.nPaths) {
    const uint2 key = W.key[i];
    alive = key.y != DEAD;
    const float nanv = __int_as_float(0x7fc00000);
    float4* ro = W.rayO + 4 * (size_t)i;
    float4* rd = W.rayD + 4 * (size_t)i;
    if (alive) {
      const float4 o4 = W.org[i], d4 = W.dir[i];
      const f3 o = mk(o4.x, o4.y, o4.z), d = mk(d4.x, d4.y, d4.z);
      f3 nrm, pt;
      uint32_t mesh;
      vertex_setup_ray(S, key.y, o, d, nrm, pt, mesh);
      Rng g{W.rng[i]};
      for (uint32_t l = 0; l < 3u; l++) {
        if (l < S.n_lights) {
          const f3 tl = light_sample(g, S.lights[l]) - pt;  // drawn unconditionally, in light order
          ro[l] = make_float4(pt.x, pt.y, pt.z, __uint_as_float(1u));  // any-hit
          rd[l] = make_float4(tl.x, tl.y, tl.z, 0.f);
        } else {
          ro[l] = make_float4(0.f, 0.f, 0.f, __uint_as_float(1u)), rd[l] = make_float4(nanv, nanv, nanv, 0.f);
        }
      }
      if (bounce) {
        const f3 bd = hemisphere_sample(g, nrm);
        ro[3] = make_float4(pt.x, pt.y, pt.z, 0.f), rd[3] = make_float4(bd.x, bd.y, bd.z, 0.f);
      } else {
        ro[3] = make_float4(0.f, 0.f, 0.f, 0.f), rd[3] = make_float4(nanv, nanv, nanv, 0.f);
      }
      W.rng[i] = g.s;
      W.nrm[i] = make_float4(nrm.x, nrm.y, nrm.z, __uint_as_float(mesh));
      W.pnt[i] = make_float4(pt.x, pt.y, pt.z, 0.f);
    } else {
      for (int k = 0; k < 4; k++)  // a NaN ray is dead on arrival (Trav::start): costs the trace stage one slot
        ro[k] = make_float4(0.f, 0.f, 0.f, __uint_as_float(k < 3 ? 1u : 0u)), rd[k] = make_float4(nanv, nanv, nanv, 0.f);
    }
  }
  const uint32_t c = wsum(alive ? 1u : 0u);
  if ((threadIdx.x & 63u) == 0 && c) {
    atomicAdd(&W.stripes[2u * (blockIdx.x & 1023u) + 1u], (unsigned long long)c * S.n_lights);
    if (bounce) atomicAdd(&W.stripes[2u * (blockIdx.x & 1023u)], (unsigned long long)c);
  }
}

__global__ __launch_bounds__(1024) void wf_sum_stripes(unsigned long long* __restrict__ stripes, unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long sh[2][16];
  unsigned long long a = stripes[2u * threadIdx.x], b = stripes[2u * threadIdx.x + 1u];
  stripes[2u * threadIdx.x] = 0, stripes[2u * threadIdx.x + 1u] = 0;  // ready for the next frame
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64), b += __shfl_xor(b, off, 64);
  if ((threadIdx.x & 63u) == 0) sh[0][threadIdx.x >> 6] = a, sh[1][threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, tb = 0;
    for (int i = 0; i < 16; i++) ta += sh[0][i], tb += sh[1][i];
    atomicAdd(&counters[RTK_CNT_CLOSEST], ta);
    atomicAdd(&counters[RTK_CNT_SHADOW], tb);
  }
}

// after the trace stage: direct lighting of the vertex, then on to the bounce hit
__global__ void wf_finish(DevScene S, WfArgs W, uint32_t depth, uint32_t bounce) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W.nPaths) return;
  const uint2 key = W.key[i];
  if (key.y == DEAD) return;
  const float4 d4 = W.dir[i], n4 = W.nrm[i], p4 = W.pnt[i];
  const f3 rayDir = mk(d4.x, d4.y, d4.z), nrm = mk(n4.x, n4.y, n4.z), pt = mk(p4.x, p4.y, p4.z);
  const uint32_t mesh = __float_as_uint(n4.w);
  const BsdfBase base = bsdf_base(S.mats[mesh], nrm, -rayDir);
  f3 color = mk(0.f, 0.f, 0.f);
  for (uint32_t l = 0; l < S.n_lights && l < 3u; l++) {
    if (W.res[4 * (size_t)i + l].x != 0u) continue;  // occluded (Renderer.cpp:54-55)
    const float4 t4 = W.rayD[4 * (size_t)i + l];
    const f3 toLight = mk(t4.x, t4.y, t4.z);
    const f3 bsdf = bsdf_apply(base, toLight);
    const f3 radiance = light_eval(S.lights[l], pt);
    color = color + radiance * bsdf;
  }
  float4& c = W.col[(size_t)depth * W.nPaths + i];
  c.x = color.x, c.y = color.y, c.z = color.z;
  if (bounce) {
    const uint2 nk = W.res[4 * (size_t)i + 3];
    const float4 b4 = W.rayD[4 * (size_t)i + 3];
    W.org[i] = make_float4(pt.x, pt.y, pt.z, 0.f);
    W.dir[i] = make_float4(b4.x, b4.y, b4.z, 0.f);
    W.key[i] = nk;  // {~0, ~0} = the bounce ray left the scene: the path ends
  } else {
    W.key[i] = make_uint2(DEAD, DEAD);
  }
}

// Renderer.cpp:254-258 for the batch: samples of a pixel are added in sample order
__global__ void wf_accumulate(WfArgs W, float4* __restrict__ accum) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;  // (granule, pixel) slot
  const uint32_t perSample = W.nGran * 64u;
  if (r >= perSample) return;
  const uint32_t g = r >> 6, l = r & 63u, gr = W.gran[g];
  const uint32_t px = (gr & 0xffffu) * 8u + (l & 7u), py = (gr >> 16) * 8u + (l >> 3);
  if (px >= W.width || py >= W.height) return;
  const uint32_t pix = py * W.width + px;
  float4 sum = accum[pix];
  const uint32_t nb = W.nPaths / perSample;
  for (uint32_t sb = 0; sb < nb && W.s0 + sb < W.s1; sb++) {
    const size_t i = (size_t)sb * perSample + r;
    const float4 c0 = W.col[i], c1 = W.col[W.nPaths + i], c2 = W.col[2 * (size_t)W.nPaths + i];
    // calculateColorPath returns c0 + (c1 + (c2 + 0))
    const f3 total = mk(c0.x, c0.y, c0.z) + (mk(c1.x, c1.y, c1.z) + (mk(c2.x, c2.y, c2.z) + mk(0.f, 0.f, 0.f)));
    sum.x += clamp01(total.x), sum.y += clamp01(total.y), sum.z += clamp01(total.z);
    if (c0.w != 0.f) sum.w += 1.f;
  }
  accum[pix] = sum;
}

hipError_t launch_trace_stream(const DevScene& S, const float4* rayO, const float4* rayD, uint32_t n, uint2* res,
                               uint32_t* counter, uint32_t stackLevels, uint32_t numCUs, hipStream_t stream);

hipError_t launch_wavefront(const DevScene& S, const WfArgs& W0, uint32_t mode, uint32_t maxDepth, float4* accum,
                            unsigned long long* counters, uint32_t* queueCounter, uint32_t stackLevels, uint32_t numCUs,
                            hipStream_t stream) {
  const uint32_t perSample = W0.nGran * 64u;
  if (perSample == 0 || W0.s1 <= W0.s0) return hipSuccess;
  const uint32_t nvert = mode == RT_MODE_PATH ? maxDepth : 1u;
  for (uint32_t s = W0.s0; s < W0.s1; s += W0.batch) {
    WfArgs W = W0;
    W.s0 = s;
    const uint32_t nb = W0.s1 - s < W0.batch ? W0.s1 - s : W0.batch;
    W.nPaths = nb * perSample;
    const dim3 blk(256), grd((W.nPaths + 255) / 256);
    hipLaunchKernelGGL(wf_generate, dim3((W.nPaths + 63) / 64), dim3(64), 0, stream, S, W, counters);
    for (uint32_t depth = 0; depth < nvert; depth++) {
      const uint32_t bounce = (mode == RT_MODE_PATH && depth + 1 < nvert) ? 1u : 0u;
      hipLaunchKernelGGL(wf_vertex, grd, blk, 0, stream, S, W, bounce, counters);
      hipError_t e = launch_trace_stream(S, W.rayO, W.rayD, 4u * W.nPaths, W.res, queueCounter, stackLevels, numCUs, stream);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(wf_finish, grd, blk, 0, stream, S, W, depth, bounce);
    }
    hipLaunchKernelGGL(wf_accumulate, dim3((perSample + 255) / 256), blk, 0, stream, W, accum);
  }
  hipLaunchKernelGGL(wf_sum_stripes, dim3(1), dim3(1024), 0, stream, W0.stripes, counters);
  return hipGetLastError();
}
