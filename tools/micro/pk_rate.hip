// Developer micro-test: issue cost of packed f32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against v_fma_f32 and
// v_mov_b32 in a plain VALU stream (no MFMA beside it): does SLP-packing two scalar f32 operations into one packed one
// save issue time once the pair-forming moves are counted?  Four waves per SIMD, eight independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_rate.bin tools/micro/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  f2 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = f2{(float)c + threadIdx.x, (float)c - threadIdx.x};
  const f2 av{a, a * 1.5f}, bv{b, b * .5f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (KIND == 0) {          // two scalar FMAs
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c].x) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c].y) : "v"(a), "v"(b));
      } else if (KIND == 1) {   // one packed FMA
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(av), "v"(bv));
      } else if (KIND == 2) {   // one packed multiply
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(av));
      } else if (KIND == 3) {   // one packed add
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(bv));
      } else if (KIND == 4) {   // two moves (what forming a pair costs)
        asm volatile("v_mov_b32 %0, %1" : "+v"(acc[c].x) : "v"(a));
        asm volatile("v_mov_b32 %0, %1" : "+v"(acc[c].y) : "v"(b));
      } else if (KIND == 5) {   // one scalar FMA
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c].x) : "v"(a), "v"(b));
      } else if (KIND == 6) {   // packed FMA with op_sel / neg modifiers (the complex-multiply form)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[c]) : "v"(av), "v"(bv));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += acc[c].x + acc[c].y;
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
#define RUN(K) timed([&](int it) { hipLaunchKernelGGL(k<K>, dim3(4096), dim3(64), 0, 0, d, it, 1.0001f, 0.5f); })
int main() {
  float* d; (void)hipMalloc(&d, 4 * 64 * 4096);
  const float t0 = RUN(0), t1 = RUN(1), t2 = RUN(2), t3 = RUN(3), t4 = RUN(4), t5 = RUN(5), t6 = RUN(6);
  printf("4 waves/SIMD, 160000 steps per wave: 2 x v_fma_f32 %.3f ms | v_pk_fma_f32 %.3f | v_pk_mul_f32 %.3f | v_pk_add_f32 %.3f | 2 x v_mov_b32 %.3f | 1 x v_fma_f32 %.3f | v_pk_fma_f32 with op_sel/neg %.3f\n",
         t0, t1, t2, t3, t4, t5, t6);
  return 0;
}
