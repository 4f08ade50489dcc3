// Developer micro-test: what do clock64() (s_memtime) and wall_clock64() (s_memrealtime) count, idle and under load?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_rates tools/micro/clock_rates.hip && /tmp/clock_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void k(long long* out, int iters, int heavy) {
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc = {0};
  half8 a = {(_Float16)1.f}, b = {(_Float16)threadIdx.x};
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (heavy == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    else if (heavy == 2) { v = fmaf(v, 1.0001f, 0.5f); v = fmaf(v, 0.9999f, 0.25f); v = fmaf(v, 1.0001f, 0.5f); v = fmaf(v, 0.9999f, 0.25f); }
    else __builtin_amdgcn_s_sleep(1);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (acc[0] + v == 12345.678f) out[2] = 1;
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[3];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"idle (s_sleep), 1 workgroup", "f16 MFMA on every SIMD", "f32 FMA chains on every SIMD"};
  for (int heavy = 0; heavy < 3; ++heavy) {
    const int iters = heavy == 0 ? 200000 : 2000000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(heavy ? 2048 : 1), dim3(256), 0, 0, d, iters, heavy);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-34s kernel %.2f ms (host events): clock64 %lld = %.1f MHz, wall_clock64 %lld = %.1f MHz\n", names[heavy], ms, h[0],
           h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
  }
  return 0;
}
