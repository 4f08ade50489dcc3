// Developer micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 (the instruction the f32 GEMM / attention kernels
// are priced against).  waves_per_simd waves per SIMD, each looping over `chains` independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_rate tools/micro/mfma_f32_rate.hip && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int wgs_per_cu) {
  const int iters = 20000, blocks = 256 * wgs_per_cu;
  float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * CHAINS * 2.0 * 32 * 32 * 2;
  printf("chains %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s\n", CHAINS, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<4>(3); run<1>(4); return 0; }
