// fd_ln.h -- the decode step's residual-stream assembly and LayerNorm as ONE piece of device code, shared by the fused step
// kernels (whisper_dec_fused.hip) and the vocabulary projection that takes the final LayerNorm in for steps of a few rows
// (whisper_dec_f16.hip): whichever kernel normalises a row, it runs these instructions, so the row's bits do not depend on it.
#pragma once
#include <hip/hip_runtime.h>

namespace crispy {

// xs = x_in + bias + part[0] + part[1] + ... (this order); part: [slice][rows][D]
template <int NP>
__device__ __forceinline__ float fd_assemble(float v, float bias, const float (&pv)[NP > 0 ? NP : 1]) {
  float x = v;
  if (NP > 0) {
    x += bias;
#pragma unroll
    for (int p = 0; p < NP; ++p) x += pv[p];
  }
  return x;
}

// One wave, one row: xn = f16(LayerNorm(xr)), a lane holds columns lane + 64 q, two passes (layernorm_h_kernel's arithmetic).
// gb: gamma [D] then beta [D] (LDS).  XLD-independent: `out` points at the row.
template <int D>
__device__ __forceinline__ void fd_layernorm_wave(const float* xr, const float* gb, _Float16* out, int lane) {
  constexpr int PER = D / 64;
  float e[PER], s = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) { e[q] = xr[lane + 64 * q]; s += e[q]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) { const float d = e[q] - mean; s2 = fmaf(d, d, s2); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
  const float rstd = 1.f / sqrtf(s2 / (float)D + 1e-5f);
#pragma unroll
  for (int q = 0; q < PER; ++q) out[lane + 64 * q] = (_Float16)((e[q] - mean) * rstd * gb[lane + 64 * q] + gb[D + lane + 64 * q]);
}

}  // namespace crispy
