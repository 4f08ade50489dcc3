// Checks the lane semantics the batched wave reductions of rn_kernels.hip rely on (v_permlane32_swap, v_permlane16_swap,
// DPP row reductions) against a host sum.  hipcc --offload-arch=gfx950 -O2 tools/micro/wave_sum4_test.hip -o /tmp/ws4 && /tmp/ws4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define RN_WAVE_SUMS_STANDALONE 1
#include "../../crispy_amd/csrc/rn_wave_sums.h"

__global__ void k(const float* in, float* out) {
  const int lane = threadIdx.x;
  float v[7];
  for (int q = 0; q < 7; ++q) v[q] = in[q * 64 + lane];
  float a[4] = {v[0], v[1], v[2], v[3]};
  crispy::wave_sums<4>(a);
  float b[2] = {v[4], v[5]};
  crispy::wave_sums<2>(b);
  float c[7] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6]};
  crispy::wave_sums<7>(c);
  if (lane == 17) {
    for (int q = 0; q < 4; ++q) out[q] = a[q];
    out[4] = b[0]; out[5] = b[1];
    for (int q = 0; q < 7; ++q) out[6 + q] = c[q];
  }
}

int main() {
  float h[7 * 64], *d_in, *d_out, o[13];
  double ref[7] = {0};
  for (int q = 0; q < 7; ++q)
    for (int l = 0; l < 64; ++l) { h[q * 64 + l] = (float)((q + 1) * 1000 + l * (q + 3)) * 0.25f; ref[q] += h[q * 64 + l]; }
  hipMalloc(&d_in, sizeof h); hipMalloc(&d_out, sizeof o);
  hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
  hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  const int idx[13] = {0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6};
  for (int i = 0; i < 13; ++i) {
    const bool ok = std::fabs(o[i] - ref[idx[i]]) <= 1e-6 * std::fabs(ref[idx[i]]);
    if (!ok) ++bad;
    printf("%2d got %.3f want %.3f %s\n", i, o[i], ref[idx[i]], ok ? "ok" : "WRONG");
  }
  printf(bad ? "FAILED\n" : "wave_sums ok\n");
  return bad != 0;
}
