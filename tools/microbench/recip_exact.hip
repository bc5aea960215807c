// recip_exact.hip — is a short reciprocal BIT-IDENTICAL to the IEEE division 1.0f / x on gfx950?
// Exhaustive over all 2^32 float bit patterns: variant A = v_rcp_f32 + one Newton step (3 instructions),
// variant B = A + a residual correction (5 instructions), reference = the compiler's correctly rounded
// expansion of 1.0f / x (11 instructions: div_scale x 2, rcp, 5 fma, div_fmas, div_fixup).  Reports mismatches
// per input range.  Not part of the product; built by `make bin/recip_exact`.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                     \
  do {                                                            \
    hipError_t e_ = (x);                                          \
    if (e_ != hipSuccess) {                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
      exit(1);                                                    \
    }                                                             \
  } while (0)

__device__ __forceinline__ float recipA(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float recipB(float x) {
  const float r = recipA(x);
  const float rem = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(rem, r, r);
}

// counts[0..2]: mismatches of A in (normal |x| in [2^-100, 2^100]) / (other finite normal) / (zero, denormal, inf, nan);
// counts[3..5]: the same for B; counts[6]: first mismatching pattern of A in the middle range (or 0)
__global__ void k_check(unsigned long long* counts, uint32_t hiBits) {
  const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;  // 2^24 threads x 256 values of the top byte... see main
  const uint32_t bits = (hiBits << 24) | lo;
  const float x = __uint_as_float(bits);
  const float ref = 1.0f / x;
  const float a = recipA(x), b = recipB(x);
  const uint32_t ex = (bits >> 23) & 255u;
  const int cls = (ex == 0 || ex == 255) ? 2 : (ex >= 27 && ex <= 227) ? 0 : 1;
  const bool nanRef = ref != ref;
  const bool badA = nanRef ? !(a != a) : __float_as_uint(a) != __float_as_uint(ref);
  const bool badB = nanRef ? !(b != b) : __float_as_uint(b) != __float_as_uint(ref);
  if (badA) {
    atomicAdd(&counts[cls], 1ull);
    if (cls == 0) atomicCAS(&counts[6], 0ull, (unsigned long long)bits);
  }
  if (badB) atomicAdd(&counts[3 + cls], 1ull);
}

// ---- square root: __builtin_sqrtf (the correctly rounded expansion, ~16 instructions) against short forms
__device__ __forceinline__ float sqrtV2(float x) {  // v_sqrt_f32 + residual correction through 0.5 / s
  const float s = __builtin_amdgcn_sqrtf(x);
  const float r = __builtin_fmaf(-s, s, x);
  const float h = 0.5f * __builtin_amdgcn_rcpf(s);
  return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrtV3(float x) {  // v_rsq_f32: s = x y, h = y / 2, one correction
  const float y = __builtin_amdgcn_rsqf(x);
  const float s = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrtV4(float x) {  // V3 + a second correction
  const float y = __builtin_amdgcn_rsqf(x);
  float s = x * y;
  const float h = 0.5f * y;
  float r = __builtin_fmaf(-s, s, x);
  s = __builtin_fmaf(r, h, s);
  r = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(r, h, s);
}
// counts[8 + 4 v + cls]: v = 0 bare v_sqrt_f32, 1 V2, 2 V3, 3 V4; cls = 0: 2^-100 <= x < 2^101, 1: other positive normal, 2: the rest
__global__ void k_check_sqrt(unsigned long long* counts, uint32_t hiBits) {
  const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t bits = (hiBits << 24) | lo;
  const float x = __uint_as_float(bits);
  const float ref = __builtin_sqrtf(x);
  const uint32_t ex = (bits >> 23) & 255u;
  const int cls = (bits >> 31) || ex == 0 || ex == 255 ? 2 : (ex >= 27 && ex <= 227) ? 0 : 1;
  const float v[4] = {__builtin_amdgcn_sqrtf(x), sqrtV2(x), sqrtV3(x), sqrtV4(x)};
  for (int k = 0; k < 4; ++k) {
    const bool bad = ref != ref ? !(v[k] != v[k]) : __float_as_uint(v[k]) != __float_as_uint(ref);
    if (bad) atomicAdd(&counts[8 + 4 * k + cls], 1ull);
  }
}

int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 32 * sizeof(unsigned long long)));
  CK(hipMemset(d, 0, 32 * sizeof(unsigned long long)));
  for (uint32_t hi = 0; hi < 256; ++hi) hipLaunchKernelGGL(k_check, dim3(1u << 16), dim3(256), 0, nullptr, d, hi);
  for (uint32_t hi = 0; hi < 256; ++hi) hipLaunchKernelGGL(k_check_sqrt, dim3(1u << 16), dim3(256), 0, nullptr, d, hi);
  CK(hipDeviceSynchronize());
  unsigned long long h[32];
  CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  printf("{\"inputs\": 4294967296, \"A_rcp_plus_one_newton_step\": {\"mismatch_exp_-100_to_100\": %llu, \"mismatch_other_normal\": %llu, "
         "\"mismatch_zero_denormal_inf_nan\": %llu, \"first_mismatch_bits\": \"0x%08llx\"}, "
         "\"B_plus_residual_correction\": {\"mismatch_exp_-100_to_100\": %llu, \"mismatch_other_normal\": %llu, \"mismatch_zero_denormal_inf_nan\": %llu}}\n",
         h[0], h[1], h[2], h[6], h[3], h[4], h[5]);
  const char* names[4] = {"bare_v_sqrt_f32", "sqrt_then_residual_times_half_rcp", "rsq_one_correction", "rsq_two_corrections"};
  printf("{\"sqrt_vs_correctly_rounded_sqrtf\": {");
  for (int k = 0; k < 4; ++k)
    printf("%s\"%s\": {\"mismatch_exp_-100_to_100\": %llu, \"mismatch_other_positive_normal\": %llu, \"mismatch_rest\": %llu}", k ? ", " : "", names[k],
           h[8 + 4 * k], h[8 + 4 * k + 1], h[8 + 4 * k + 2]);
  printf("}}\n");
  return 0;
}
