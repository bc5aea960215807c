// exec_half.hip — does a wave64 VALU instruction cost less when one half of EXEC is empty?
// Three masks over the same VALU-bound loop (16 waves per CU, 4 per SIMD): all 64 lanes, the low 32,
// 32 scattered lanes (every other one).  Prints ms per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63;
  bool on = true;
  if (MODE == 1) on = lane < 32;
  if (MODE == 2) on = (lane & 1) == 0;
  if (MODE == 3) on = lane >= 32;
  if (MODE == 4) on = lane < 16;
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  if (on) {
    for (int i = 0; i < iters; ++i) {
      x0 = __builtin_fmaf(x0, a, b), x1 = __builtin_fmaf(x1, a, b), x2 = __builtin_fmaf(x2, a, b), x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b), x5 = __builtin_fmaf(x5, a, b), x6 = __builtin_fmaf(x6, a, b), x7 = __builtin_fmaf(x7, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
static float run(float* out, int cus, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(cus), dim3(1024), 0, nullptr, out, 64, 1.0001f, 0.5f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(cus), dim3(1024), 0, nullptr, out, iters, 1.0001f, 0.5f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float* out;
  CK(hipMalloc(&out, (size_t)cus * 1024 * 4));
  const int iters = 200000;
  printf("{\"all64_ms\": %.3f, ", run<0>(out, cus, iters));
  printf("\"low32_ms\": %.3f, ", run<1>(out, cus, iters));
  printf("\"even32_ms\": %.3f, ", run<2>(out, cus, iters));
  printf("\"high32_ms\": %.3f, ", run<3>(out, cus, iters));
  printf("\"low16_ms\": %.3f, \"valu_per_wave\": %lld}\n", run<4>(out, cus, iters), 8ll * iters);
  return 0;
}
