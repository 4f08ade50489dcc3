// Developer micro-test: issue cost of the f32 <-> f64 conversions beside v_fma_f64 (the high-pass recurrence is made
// of three of each plus one f32 add per sample).  Four waves per SIMD, eight independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/cvt_rate.bin tools/micro/cvt_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_fma64(double* out, int iters, double a, double b) {
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (double)c + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
// one round trip f64 -> f32 -> f64 per step (two conversions)
__global__ __launch_bounds__(64) void k_cvt(double* out, int iters, double a) {
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (double)c + threadIdx.x * a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float f;
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(acc[c]));
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(acc[c]) : "v"(f));
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F>
float timed(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(100);
  (void)hipEventRecord(e0);
  f(20000);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double* d; (void)hipMalloc(&d, 8 * 64 * 4096);
  const float t1 = timed([&](int it) { hipLaunchKernelGGL(k_fma64, dim3(4096), dim3(64), 0, 0, d, it, 1.0001, 0.5); });
  const float t2 = timed([&](int it) { hipLaunchKernelGGL(k_cvt, dim3(4096), dim3(64), 0, 0, d, it, 0.37); });
  printf("4 waves/SIMD: 160000 v_fma_f64 per wave %.3f ms; 160000 x (v_cvt_f32_f64 + v_cvt_f64_f32) per wave %.3f ms\n", t1, t2);
  return 0;
}
