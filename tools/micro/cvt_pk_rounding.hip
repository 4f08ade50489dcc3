// Developer micro-test: does v_cvt_pk_f16_f32 round like v_cvt_f16_f32 (to nearest even)?  Both over 1 M random floats.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/cvt_pk tools/micro/cvt_pk_rounding.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void k(const float* x, uint16_t* a, uint16_t* b, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const _Float16 ha = (_Float16)x[i];
  unsigned pk;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[i]), "v"(x[i]));
  __builtin_memcpy(&a[i], &ha, 2);
  b[i] = (uint16_t)(pk & 0xffffu);
}
int main() {
  const int n = 1 << 20;
  float* hx = new float[n];
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * (i % 3 == 0 ? 1e-3f : i % 3 == 1 ? 4.f : 300.f);
  float* dx; uint16_t *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
  uint16_t* ha = new uint16_t[n]; uint16_t* hb = new uint16_t[n];
  hipMemcpy(ha, da, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 2, hipMemcpyDeviceToHost);
  int diff = 0;
  for (int i = 0; i < n; ++i) diff += ha[i] != hb[i];
  printf("v_cvt_pk_f16_f32 against v_cvt_f16_f32: %d of %d results differ\n", diff, n);
  return 0;
}
