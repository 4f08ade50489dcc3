// Register layout of v_mfma_f32_16x16x32_f16 as the fused decode kernels use it (A = 16 weight rows x 32 k, B = 32 k x
// 16 batch rows): checks D[m][n] = sum_k A[m][k] B[k][n] with  A: lane l holds row l & 15, k = 8 (l >> 4) .. + 7;
// B: lane l holds column l & 15, k = 8 (l >> 4) .. + 7;  D: lane l holds column l & 15, rows 4 (l >> 4) .. + 3.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma16_layout.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* A, const _Float16* B, float* D) {      // A [16][32], B [16 cols][32 k] (column-major rows), D [16][16]
  const int l = threadIdx.x;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = B[(l & 15) * 32 + 8 * (l >> 4) + e]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}
int main() {
  _Float16 hA[16 * 32], hB[16 * 32];
  float hD[256];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 20) % 17 - 8) / 4.f; };
  for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)rnd(); hB[i] = (_Float16)rnd(); }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double r = 0;
      for (int kk = 0; kk < 32; ++kk) r += (double)hA[m * 32 + kk] * (double)hB[n * 32 + kk];
      worst = fmax(worst, fabs(r - hD[m * 16 + n]));
    }
  printf("mfma_f32_16x16x32_f16 layout check: worst |diff| %.3g (%s)\n", worst, worst < 1e-3 ? "OK" : "WRONG");
  return worst < 1e-3 ? 0 : 1;
}
