// rt_device.h — device-side building blocks of the gfx950 path integrator.
//
// Each function names the reference lines whose RESULT it must reproduce; the
// code is written for the GPU (flattened records, no recursion, no containers).
// Bit-parity rules (SURVEY.md §7 "hard parts"): this translation unit is built
// with -ffp-contract=off, every expression keeps the reference's operand types
// and association, division and sqrt are the correctly rounded HIP defaults,
// and the transcendental functions come from include/rt_pixelmode.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"
#include "rt_pixelmode.h"

#define RT_DEV __device__ __forceinline__

namespace rtd {

// ------------------------------------------------------------------ 3-vectors
struct f3 {
  float x, y, z;
};
RT_DEV f3 mk(float x, float y, float z) { return f3{x, y, z}; }
RT_DEV f3 ld(const float* p) { return f3{p[0], p[1], p[2]}; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator*(float s, f3 a) { return mk(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// Vec3.h:221-232 — products summed left to right, cross as written there
RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross3(f3 a, f3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// NB: __fsqrt_rn() lowers to a bare v_sqrt_f32 (1 ulp) on gfx950; sqrtf / __builtin_sqrtf
// get the correctly rounded expansion (v_sqrt_f32 + residual fix-up).
RT_DEV float len3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
// 1.0f / x in three instructions (v_rcp_f32 + one Newton step) instead of the eleven of the correctly rounded
// expansion — and the SAME BITS for every x with 2^-100 <= |x| < 2^101: checked exhaustively on gfx950 over all 2^32
// inputs (tools/microbench/recip_exact.hip, profiles/r03_recip_exact.json: 0 mismatches in that range; outside it —
// results in or near the denormal range, zero, infinity, NaN — the two differ, so every caller guards its range and
// takes the division there).
RT_DEV float recip_fast(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
// sqrtf(x) in five instructions (v_rsq_f32, s = x y, h = y / 2, one residual correction) instead of the sixteen of the
// correctly rounded expansion, and the SAME BITS for every x with 2^-100 <= x < 2^101 (the same exhaustive check: 0
// mismatches; v_sqrt_f32 alone differs on 255 M inputs of that range, v_sqrt_f32 + correction on 100).
RT_DEV float sqrt_fast(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s = x * y, h = 0.5f * y;
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
constexpr float kRecipLo = 7.888609e-31f;  // 2^-100
constexpr float kRecipHi = 1.2676506e30f;  // 2^100
// Vec3.h:170-178 — null vectors stay null, otherwise multiply by 1/len.
// FAST (a compile-time choice of the kernel instance, rt_kernels.hip LT_FASTDET): the length by sqrt_fast and 1 / length by
// recip_fast where the squared length is in [2^-100, 2^100) — the upper end is the host's promise for the instances that
// say FAST (rt_api.cpp create_ctx bounds every input of the scene by 1e14), the lower end is tested here; the length is
// then in [2^-50, 2^50), inside recip_fast's range.  Same bits as sqrtf and the division.
// (the general path of the FAST form, out of line: eleven inlined copies of the sqrtf + division expansions that are never
// executed cost the big-scene instances 1 % through the instruction cache)
__device__ __attribute__((noinline)) inline f3 unit3_general(f3 a) {
  float l = len3(a);
  if (l == 0.f) return a;
  const float inv = 1.0f / l;
  return mk(a.x * inv, a.y * inv, a.z * inv);
}
template <bool FAST = false>
RT_DEV f3 unit3(f3 a) {
  if (FAST) {
    const float dd = dot3(a, a);
    // (dd < 2^100 is the host's bound; below 2^-100 — a null vector, a denormal square, a NaN — the general path)
    if (dd >= kRecipLo) {
      const float inv = recip_fast(sqrt_fast(dd));
      return mk(a.x * inv, a.y * inv, a.z * inv);
    }
    return unit3_general(a);
  }
  float l = len3(a);
  if (l == 0.f) return a;
  const float inv = 1.0f / l;
  return mk(a.x * inv, a.y * inv, a.z * inv);
}
RT_DEV float dist3(f3 a, f3 b) { return len3(a - b); }

// ------------------------------------------------------------------ RNG stream
// minstd_rand0 (LightSource.h:6) + libstdc++ generate_canonical semantics
// (SURVEY.md App. B): float draw = 1 engine call, double draw = 2 calls.
struct Rng {
  uint32_t s;
  RT_DEV uint32_t next() {
    uint64_t p = (uint64_t)s * 16807ull;                       // < 2^46
    uint32_t x = (uint32_t)(p & 0x7fffffffull) + (uint32_t)(p >> 31);
    if (x >= 0x7fffffffu) x -= 0x7fffffffu;                    // mod 2^31-1
    s = x;
    return x;
  }
  RT_DEV float canonF() {
    float r = (float)(next() - 1u) / 2147483648.0f;
    return r >= 1.0f ? 0.99999994f : r;
  }
  RT_DEV double canonD() {
    const double R = 2147483646.0, R2 = 2147483646.0 * 2147483646.0;
    double sum = (double)(next() - 1u);
    sum += (double)(next() - 1u) * R;
    double r = sum / R2;
    return r >= 1.0 ? 0.99999999999999988898 : r;
  }
  RT_DEV float uniformF(float a, float b) { return canonF() * (b - a) + a; }
  RT_DEV double uniformD(double a, double b) { return canonD() * (b - a) + a; }
};

// ------------------------------------------------------------------ samplers
// RayTracer.h:109-117
RT_DEV void jitter_sample(Rng& g, int idx, int n, float& x, float& y) {
  int d = (int)__builtin_sqrtf((float)n);
  int j2 = idx / d, i2 = idx % d;
  x = (float)(((double)(float)i2 + g.uniformD(0.0, 1.0)) / (double)(float)d);
  y = (float)(((double)(float)j2 + g.uniformD(0.0, 1.0)) / (double)(float)d);
}

// Vec3.h:180-199
RT_DEV void two_orthogonals(f3 n, f3& u, f3& v) {
  float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
  if (ax < ay) u = (ax < az) ? mk(0.f, -n.z, n.y) : mk(-n.y, n.x, 0.f);
  else u = (ay < az) ? mk(n.z, 0.f, -n.x) : mk(-n.y, n.x, 0.f);
  v = cross3(n, u);
}

// RayTracer.h:95-107 with maxRayAngle = float(pi/2) (its only call value)
template <bool FAST = false>
RT_DEV f3 hemisphere_sample(Rng& g, f3 normal) {
  const float maxRayAngle = 1.57079637f;
  const double PI = 3.14159265358979323846;
  const double hi = (double)(2 * maxRayAngle) / PI;
  normal = unit3<FAST>(normal);
  f3 v1, v2;
  two_orthogonals(normal, v1, v2);
  v1 = unit3<FAST>(v1);
  v2 = unit3<FAST>(v2);
  float theta = (float)rt_asin(g.uniformD(0.0, hi));
  float phi = (float)(2 * PI * g.uniformD(0.0, hi));
  float sp, cp, st, ct;
  rt_sincosf(phi, &sp, &cp);
  rt_sincosf(theta, &st, &ct);
  f3 dir = unit3<FAST>(v1 * cp + v2 * sp);
  return unit3<FAST>(normal * ct + dir * st);
}

// ------------------------------------------------------------------ camera / lights
// Camera.h:27-30
template <bool FAST = false>
RT_DEV void camera_ray(const rt_camera& c, float u, float v, f3& o, f3& d) {
  o = ld(c.position);
  d = unit3<FAST>(ld(c.lower_left) + u * ld(c.horizontal) + v * ld(c.vertical) - o);
}

// LightSource.h:46-49; first draw scales the horizontal axis (g++ evaluation order)
// (in two halves, so that a caller may keep the two parameters and form the point later)
RT_DEV void light_sample_params(Rng& g, const rt_light& l, float& rh, float& rv) {
  rh = g.uniformF(-l.side, l.side);
  rv = g.uniformF(-l.side, l.side);
}
RT_DEV f3 light_point(const rt_light& l, float rh, float rv) {
  return ld(l.position) + (rv * ld(l.vertical)) + (rh * ld(l.horizontal));
}
RT_DEV f3 light_sample(Rng& g, const rt_light& l) {
  float rh, rv;
  light_sample_params(g, l, rh, rv);
  return light_point(l, rh, rv);
}
// LightSource.h:51-54
RT_DEV float light_radiance(const rt_light& l, f3 p) {
  float d = dist3(p, ld(l.position));
  return l.intensity / (l.ac + l.al * d + l.aq * d * d);
}
// LightSource.h:56-59
RT_DEV f3 light_eval(const rt_light& l, f3 p) { return l.factor * ld(l.color) * light_radiance(l, p); }

// ------------------------------------------------------------------ material
// Material.h:25-70.  Mixed precision exactly as there: D's denominator, the two
// pow(), fmax(0,.), 4.*(n.wi)*(n.wo) and sqrt(2/pi) are double, each narrowed to
// float where the reference assigns to a float.
RT_DEV float g_schlick(float alpha, f3 w, f3 n) {
  float k = (float)((double)alpha * 0x1.9884533d43651p-1 /* == sqrt(2. / M_PI) in double */);
  float nw = dot3(n, w);
  return nw / (nw * (1 - k) + k);
}
// What Material.h:25-70 computes from the material ALONE, hoisted to rt_create (rt_api.cpp fills it with the same IEEE
// operations on the same operands: albedo / float(M_PI) then * kd, 1 - kd, alpha * alpha, 1 - F0, k = float(double(alpha) *
// sqrt(2 / pi)), 1 - k — the bits the per-vertex code formed): 3 divisions, a double product and a dozen other operations
// per vertex and four more double products per vertex in gSchlick that no longer run in the pooled kernel.
struct DevMat {
  float kdDiffuse[3], oneMinusKd;
  float f0[3], alpha;
  float oneMinusF0[3], a2;
  float k, oneMinusK, pad[2];
};
static_assert(sizeof(DevMat) == 64, "device material record");
inline DevMat make_dev_mat(const rt_material& m) {  // (host side, rt_create)
  DevMat d;
  for (int c = 0; c < 3; ++c) {
    const float diffuse = m.albedo[c] / 3.14159274f;  // albedo / float(M_PI)
    d.kdDiffuse[c] = m.kd * diffuse;
    d.f0[c] = m.f0[c];
    d.oneMinusF0[c] = 1.f - m.f0[c];
  }
  d.oneMinusKd = 1 - m.kd;
  d.alpha = m.alpha, d.a2 = m.alpha * m.alpha;
  d.k = (float)((double)m.alpha * 0x1.9884533d43651p-1 /* == sqrt(2. / M_PI) in double */);
  d.oneMinusK = 1 - d.k;
  d.pad[0] = d.pad[1] = 0.f;
  return d;
}
RT_DEV float g_schlick_k(float k, float oneMinusK, f3 w, f3 n) {
  float nw = dot3(n, w);
  return nw / (nw * oneMinusK + k);
}
// evaluateColorResponse split in two: what depends on (material, normal, wo) only is
// computed once per vertex, the rest once per light.  Same operations on the same
// operands in the same order as the one-piece form, so the same bits.
struct BsdfBase {
  f3 n, wo, F0, oneMinusF0, kdDiffuse;
  float alpha, a2, gwo, nwo, oneMinusKd;
  float k, oneMinusK;  // gSchlick's constants (set by both constructors)
};
template <bool FAST = false>
RT_DEV BsdfBase bsdf_base(const rt_material& m, f3 normal, f3 wo_in) {
  BsdfBase B;
  B.n = unit3<FAST>(normal), B.wo = unit3<FAST>(wo_in);
  B.alpha = m.alpha, B.a2 = m.alpha * m.alpha;
  B.F0 = ld(m.f0), B.oneMinusF0 = mk(1.f, 1.f, 1.f) - B.F0;
  B.gwo = g_schlick(m.alpha, B.wo, B.n);
  B.nwo = dot3(B.n, B.wo);
  const f3 diffuse = ld(m.albedo) / 3.14159274f;  // albedo / float(M_PI)
  B.kdDiffuse = m.kd * diffuse, B.oneMinusKd = 1 - m.kd;
  B.k = (float)((double)m.alpha * 0x1.9884533d43651p-1), B.oneMinusK = 1 - B.k;
  return B;
}
// the same from the hoisted record
template <bool FAST = false>
RT_DEV BsdfBase bsdf_base(const DevMat& m, f3 normal, f3 wo_in) {
  BsdfBase B;
  B.n = unit3<FAST>(normal), B.wo = unit3<FAST>(wo_in);
  B.alpha = m.alpha, B.a2 = m.a2;
  B.F0 = ld(m.f0), B.oneMinusF0 = ld(m.oneMinusF0);
  B.k = m.k, B.oneMinusK = m.oneMinusK;
  B.gwo = g_schlick_k(B.k, B.oneMinusK, B.wo, B.n);
  B.nwo = dot3(B.n, B.wo);
  B.kdDiffuse = ld(m.kdDiffuse), B.oneMinusKd = m.oneMinusKd;
  return B;
}
template <bool FAST = false>
RT_DEV f3 bsdf_apply(const BsdfBase& B, f3 wi_in) {
  const double PI = 3.14159265358979323846;
  f3 wi = unit3<FAST>(wi_in);
  f3 wh = unit3<FAST>(wi + B.wo);
  float D = (float)((double)B.a2 / (PI * rt_pow2(1 + (double)(B.a2 - 1) * rt_pow2((double)dot3(B.n, wh)))));
  double c = (double)dot3(wi, wh);
  float f5 = (float)rt_pow5(1 - (c > 0.0 ? c : 0.0));  // fmax(0, c); NaN -> 0 like fmax
  f3 F = B.F0 + B.oneMinusF0 * f5;
  float G = g_schlick_k(B.k, B.oneMinusK, wi, B.n) * B.gwo;
  float denom = (float)(4. * (double)dot3(B.n, wi) * (double)B.nwo);
  f3 spec = D * F * G / denom;
  f3 r = B.kdDiffuse + B.oneMinusKd * spec;
  if (r.x < 0.f) r.x = 0.f;
  if (r.y < 0.f) r.y = 0.f;
  if (r.z < 0.f) r.z = 0.f;
  return r;
}
RT_DEV f3 bsdf_eval(const rt_material& m, f3 normal, f3 wi_in, f3 wo_in) {
  return bsdf_apply(bsdf_base(m, normal, wo_in), wi_in);
}

// ------------------------------------------------------------------ triangle test
// Ray.cpp:9-24 on a flattened record (p0, e1 = p1-p0, e2 = p2-p0 precomputed with
// the same float subtraction the reference performs per test).
// (the part of the test behind the reciprocal of the determinant)
RT_DEV bool tri_finish(f3 o, f3 d, f3 p0, f3 e1, f3 e2, f3 pvec, float det, float inv, float& u, float& v, float& t) {
  f3 tvec = o - p0;
  u = dot3(tvec, pvec) * inv;
  f3 qvec = cross3(tvec, e1);
  v = dot3(d, qvec) * inv;
  t = dot3(e2, qvec) * inv;
  const bool detOk = !(fabsf(det) < 0.000001f);
  const bool uOk = !(u < 0.f || u > 1.f);
  return detOk && uOk && v >= 0.f && u + v <= 1.f;
}
// `divide` (a compile-time constant at every call site) = this kernel instance's rays are not known to keep |det| below
// 2^100 (rt_kernels.hip LT_FASTDET): 1 / det by the division.
RT_DEV bool tri_test(f3 o, f3 d, f3 p0, f3 e1, f3 e2, float& u, float& v, float& t, bool divide) {
  // Branch-free form: the reference returns early when |det| < EPSILON or u is out of
  // range; here everything is evaluated (a zero det just yields inf/NaN values that
  // the combined predicate discards) so that a wave never splits inside the test.
  // The RESULT (bool and, when true, u v t) is the reference's bit for bit.
  f3 pvec = cross3(d, e2);
  float det = dot3(e1, pvec);
  // 1 / det: three instructions where the host has bounded |det| for every ray this launch can produce (rt_api.cpp
  // create_ctx: longest edge^2 x longest ray direction < 2^100; a determinant below recip_fast's range is below
  // EPSILON too and rejected whatever inv holds), the division otherwise.  Same bits either way.
  const float inv = divide ? 1.0f / det : recip_fast(det);
  return tri_finish(o, d, p0, e1, e2, pvec, det, inv, u, v, t);
}
// Same test with the reference's early exits (leaves u, v, t untouched exactly where
// Ray.cpp:9-24 does) — the unit-test hook compares those side effects too.
RT_DEV bool tri_test_ref_order(f3 o, f3 d, f3 p0, f3 e1, f3 e2, float& u, float& v, float& t) {
  f3 pvec = cross3(d, e2);
  float det = dot3(e1, pvec);
  if (fabsf(det) < 0.000001f) return false;
  float inv = 1.0f / det;
  f3 tvec = o - p0;
  u = dot3(tvec, pvec) * inv;
  f3 qvec = cross3(tvec, e1);
  v = dot3(d, qvec) * inv;
  t = dot3(e2, qvec) * inv;
  if (u < 0.f || u > 1.f) return false;
  return v >= 0.f && u + v <= 1.f;
}

// Renderer.cpp:279-283; fminf/fmaxf return the non-NaN operand, so NaN -> 1
RT_DEV float clamp01(float c) { return fmaxf(fminf(c, 1.f), 0.f); }

}  // namespace rtd
