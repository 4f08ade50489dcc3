// Developer micro-test: does a VALU instruction cost less when only one 16-lane row of the wave is active?
// Four waves per SIMD run a dependent-free stream of v_fma_f64 / v_fma_f32 with 64, 32 or 16 active lanes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_row_skip.bin tools/micro/valu_row_skip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ __launch_bounds__(64) void k(T* out, int iters, int active, T a, T b) {
  if ((int)threadIdx.x >= active) return;
  T acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (T)c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  T s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class T>
void run(const char* name) {
  T* d; (void)hipMalloc(&d, sizeof(T) * 64 * 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int active : {64, 32, 16, 1}) {
    hipLaunchKernelGGL(k<T>, dim3(4096), dim3(64), 0, 0, d, 100, active, (T)1.0001, (T)0.5);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(4096), dim3(64), 0, 0, d, 20000, active, (T)1.0001, (T)0.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %2d active lanes, 4 waves/SIMD x 160000 fma: %.3f ms\n", name, active, ms);
  }
  (void)hipFree(d);
}
int main() { run<float>("v_fma_f32"); run<double>("v_fma_f64"); return 0; }
