/* rt_pixelmode.h — the "pixel RNG mode" determinism contract.
 *
 * The reference draws every random number from ONE global, sequential
 * std::minstd_rand0 (reference source/LightSource.h:6) in pixel-scan order,
 * which no parallel renderer can reproduce.  Pixel mode keeps
 * the reference's engine, its distributions and its draw ORDER inside one
 * sample but re-seeds the engine per (pixel, sample) — and
 * per photon for emission — from the key below.  It also pins the three
 * transcendental functions the integrator needs (asin in double, sinf, cosf;
 * reference source/RayTracer.h:102-106) and the two integer powers of the BSDF
 * (reference source/Material.h:46-52) to sequences of IEEE-754 + - * / sqrt
 * operations, so that a host CPU and a gfx950 GPU produce identical bits when
 * both are compiled without FMA contraction (-ffp-contract=off).
 *
 * This header is the whole contract; it is included by the HIP device code,
 * by the host layer and by the CPU oracle.  It contains no reference code.
 *
 * asin/sin/cos polynomial coefficients are the classic fdlibm minimax sets:
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunPro, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this software is freely
 *   granted, provided that this notice is preserved.
 */
#ifndef RT_PIXELMODE_H
#define RT_PIXELMODE_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD static inline
#endif

/* stream domains */
#define RT_STREAM_PIXEL 0u  /* index = y*width + x, sub = sample i           */
#define RT_STREAM_PHOTON 1u /* index = light*photonsPerLight + j, sub = 0    */

RT_HD uint64_t rt_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}

/* Initial minstd_rand0 state in [1, 2147483646] for one stream. */
RT_HD uint32_t rt_stream_seed(uint32_t seed, uint32_t domain, uint32_t index,
                              uint32_t sub) {
  uint64_t h = rt_mix64(((uint64_t)seed << 32) | (uint64_t)domain);
  h = rt_mix64(h + (((uint64_t)index << 32) | (uint64_t)sub));
  uint32_t s = (uint32_t)(h >> 33); /* 31 bits */
  if (s >= 2147483646u) s -= 2147483646u;
  return s + 1u;
}

/* ---- bit helpers ---------------------------------------------------------- */
RT_HD uint64_t rt_d2u(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
RT_HD double rt_u2d(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

/* ---- integer powers (pixel mode replaces libm pow(x,2), pow(x,5)) --------- */
RT_HD double rt_pow2(double x) { return x * x; }
RT_HD double rt_pow5(double x) {
  double x2 = x * x;
  return (x2 * x2) * x;
}

/* ---- sqrt that is correctly rounded on both sides -------------------------- */
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD double rt_sqrt_d(double x) { return __dsqrt_rn(x); }
#else
RT_HD double rt_sqrt_d(double x) { return __builtin_sqrt(x); }
#endif

/* ---- asin on [-1,1] in double; |x|>1 -> NaN -------------------------------- */
RT_HD double rt_asin_poly(double t) { /* t = x*x, returns (asin(x)-x)/x^3 * ... as p/q */
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  return p / q;
}

RT_HD double rt_asin(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
               pio4_hi = 7.85398163397448278999e-01;
  double ax = x < 0.0 ? -x : x;
  if (!(ax <= 1.0)) return rt_u2d(0x7ff8000000000000ull); /* NaN, also for NaN input */
  if (ax == 1.0) return x * pio2_hi + x * pio2_lo;
  if (ax < 0.5) {
    if (ax < 7.450580596923828125e-09) return x; /* 2^-27 */
    return x + x * rt_asin_poly(x * x);
  }
  double w = 1.0 - ax;
  double t = w * 0.5;
  double r = rt_asin_poly(t);
  double s = rt_sqrt_d(t);
  double res;
  if (ax >= 0.975) {
    res = pio2_hi - (2.0 * (s + s * r) - pio2_lo);
  } else {
    double sh = rt_u2d(rt_d2u(s) & 0xffffffff00000000ull);
    double c = (t - sh * sh) / (s + sh);
    double p = 2.0 * s * r - (pio2_lo - 2.0 * c);
    double q = pio4_hi - 2.0 * sh;
    res = pio4_hi - (p - q);
  }
  return x < 0.0 ? -res : res;
}

/* ---- sinf / cosf for finite |x| < ~1e5, evaluated in double ---------------- */
RT_HD double rt_ksin(double r) { /* |r| <= pi/4 */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double v = z * r;
  double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return r + v * (S1 + z * p);
}
RT_HD double rt_kcos(double r) { /* |r| <= pi/4 */
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  return 1.0 - (0.5 * z - z * p);
}
RT_HD double rt_reduce_pio2(float xf, int* quadrant) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11;
  double x = (double)xf;
  double fn = x * invpio2;
  /* round to nearest integer without libm: valid for |fn| < 2^31 */
  int n = (int)(fn < 0.0 ? fn - 0.5 : fn + 0.5);
  double dn = (double)n;
  double r = (x - dn * pio2_1) - dn * pio2_1t;
  *quadrant = n & 3;
  return r;
}
RT_HD float rt_sinf(float x) {
  int q;
  double r = rt_reduce_pio2(x, &q);
  double v;
  if (q == 0) v = rt_ksin(r);
  else if (q == 1) v = rt_kcos(r);
  else if (q == 2) v = -rt_ksin(r);
  else v = -rt_kcos(r);
  return (float)v;
}
RT_HD float rt_cosf(float x) {
  int q;
  double r = rt_reduce_pio2(x, &q);
  double v;
  if (q == 0) v = rt_kcos(r);
  else if (q == 1) v = -rt_ksin(r);
  else if (q == 2) v = -rt_kcos(r);
  else v = rt_ksin(r);
  return (float)v;
}

/* Both at once: one argument reduction, each kernel polynomial evaluated once, results
 * selected by quadrant — the same operations on the same operands as rt_sinf(x) and
 * rt_cosf(x), so the same bits (the device uses this form: no divergent quadrant
 * branches, half the double-precision work). */
RT_HD void rt_sincosf(float x, float* s, float* c) {
  int q;
  double r = rt_reduce_pio2(x, &q);
  double ks = rt_ksin(r), kc = rt_kcos(r);
  double sv = (q & 1) ? kc : ks, cv = (q & 1) ? ks : kc;
  if (q & 2) sv = -sv;
  if (q == 1 || q == 2) cv = -cv;
  *s = (float)sv;
  *c = (float)cv;
}

#endif /* RT_PIXELMODE_H */
