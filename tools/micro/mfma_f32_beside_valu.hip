// Developer micro-benchmark: does v_mfma_f32_16x16x4_f32 leave the vector ALU free?
// Workgroups of 512 threads = two waves per SIMD.  mode 0: both waves run a chain of dependent f32 MFMAs; mode 1: both run
// independent v_fma_f32 chains; mode 2: the even waves run the MFMA loop, the odd waves the FMA loop (one of each per SIMD);
// modes 3 / 4: the MFMA (FMA) wave of mode 2 alone, its partner exits at once.  If the two pipes are independent, mode 2
// takes max(mode 3, mode 4); if the f32-input matrix instruction is executed on the vector ALU, it takes their sum.
// The same for v_mfma_f32_32x32x16_f16 (modes 5 - 7) as the control: that one is known to run beside the VALU.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_beside tools/micro/mfma_f32_beside_valu.hip && gpurun_out/mfma_beside
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float mfma_f32_loop(int iters, float a, float b) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
    }
  }
  return acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}
__device__ __forceinline__ float mfma_f16_loop(int iters, float a, float b) {
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  half8 ha, hb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)a; hb[e] = (_Float16)b; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  return s;
}
__device__ __forceinline__ float fma_loop(int iters, float a, float b) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = a + e;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], b, a);      // 128 v_fma_f32 per trip, eight independent chains
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  return s;
}

__global__ __launch_bounds__(512) void k(float* out, int mode, int it_m, int it_v, float a, float b) {
  const int wave = threadIdx.x >> 6;          // waves w and w + 4 share a SIMD
  const bool first = wave < 4;
  float s = 0.f;
  switch (mode) {
    case 0: s = mfma_f32_loop(it_m, a, b); break;
    case 1: s = fma_loop(it_v, a, b); break;
    case 2: s = first ? mfma_f32_loop(it_m, a, b) : fma_loop(it_v, a, b); break;
    case 3: if (first) s = mfma_f32_loop(it_m, a, b); break;
    case 4: if (!first) s = fma_loop(it_v, a, b); break;
    case 5: s = first ? mfma_f16_loop(it_m, a, b) : fma_loop(it_v, a, b); break;
    case 6: if (first) s = mfma_f16_loop(it_m, a, b); break;
    default: break;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* d; hipMalloc(&d, sizeof(float) * 256 * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // 16 f32 MFMAs of 32 cycles per trip = 512 cycles; 128 FMAs of 4 cycles per trip = 512 cycles: equal work per trip
  const int it_m = 20000, it_v = 20000;
  const char* names[] = {"f32 MFMA in both waves of a SIMD", "v_fma_f32 in both waves", "f32 MFMA wave beside a v_fma_f32 wave",
                         "the f32 MFMA wave alone", "the v_fma_f32 wave alone", "f16 MFMA wave beside a v_fma_f32 wave",
                         "the f16 MFMA wave alone"};
  for (int mode = 0; mode < 7; ++mode) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, 100, 100, 1.0001f, 0.9999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, it_m, it_v, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  %-42s %.3f ms\n", mode, names[mode], ms);
  }
  hipFree(d);
  return 0;
}
