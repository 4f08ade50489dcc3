// rn_highpass.hip -- the two kernels around the frame kernel of the batched RNNoise path (rn_kernels.hip):
//   rn_highpass_kernel / rn_highpass_deep_kernel   the input biquad, one LANE per stream, bit-exact with the oracle
//   rn_roll_history_kernel                         keeps the last 4 high-passed frames of every stream for the next call
// Reference: nnnoiseless::DenoiseState::process_frame's first step (audio.rs:268 call site; xiph/rnnoise denoise.c biquad).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rn_common.h"

namespace crispy {
namespace {

constexpr int WAVE = 64;
// amdgpu_num_vgpr counts the unified VGPR + AGPR file on gfx950, so the attribute wants half the number: 32 registers,
// which is what lets a high-pass wave run beside four 120-register frame waves of a SIMD
#define RN_HP_VGPR_CAP __attribute__((amdgpu_num_vgpr(16)))
constexpr int RN_HP_BLK = 2;

// =============================================================================================
// high-pass: one lane per stream, strictly sequential (Appendix A.3 step 1, double products)
// =============================================================================================
// BLK = float4 blocks per request group.  2 (8 samples, 32 registers): the form that fits beside four frame waves of a
// SIMD at thousands of streams.  8 (32 samples ahead, no register cap): few streams, where this chain is what a call
// waits for and nothing competes for registers -- in the stream-major layout every lane reads from its own 5.8 MB
// region (a TLB entry each), and one block of eight samples ahead (~0.4 us of chain) does not cover a miss:
// 1024 streams x 3001 frames 127 ms per call in that layout against 89 ms frame-major, same frame kernels.
// S16: the input is int16 PCM (RnArgs::in_s16) -- a block of four samples is 8 bytes, converted as it moves from the
// request registers to the chain (int16 -> f32 is exact); everything behind the conversion is the same instructions.
template <int BLK, bool S16>
__device__ __forceinline__ void rn_highpass_body(const RnArgs& a) {
  const int b = blockIdx.x * WAVE + threadIdx.x;
  if (b >= a.B) return;
  // One lane per stream, a chain of ten f64-path instructions per sample that nothing inside the wave can overlap: the
  // frame kernels wait for it (with few streams it IS the critical path of a call), so this wave issues ahead of the
  // frame waves that share its SIMD.
  __builtin_amdgcn_s_setprio(3);
  const double a0h = 0.5 * (double)-1.99599f, a1 = (double)0.99600f;   // b = (-2, 1) is folded into the two fmas
  float m0 = a.hp_mem[2 * b], m1 = a.hp_mem[2 * b + 1];
  float* dst = a.xhp + (long)b * a.xhp_stride + RN_HIST;
  // The recurrence is a dependent chain of five operations per sample; what the lane must not also wait for is its
  // input.  Blocks of 4 RN_HP_BLK samples are requested one block ahead, across frame boundaries: with the load
  // issued right in front of its use the kernel spent most of its time on one L1/L2 round trip per four samples.
  constexpr int NBLK = RN_FRAME / 4 / BLK;   // blocks per frame
  static_assert(RN_FRAME % (4 * BLK) == 0, "whole blocks per frame");
  const long total = (long)a.T * NBLK;
  typedef typename std::conditional<S16, short4, float4>::type Raw;      // four samples as the caller holds them
  auto block_ptr = [&](long k) {
    const long t = k / NBLK, blk = k - t * NBLK;
    const long at = t * a.stride_t + (long)b * a.stride_b;
    const Raw* p = S16 ? reinterpret_cast<const Raw*>(reinterpret_cast<const int16_t*>(a.in) + at) : reinterpret_cast<const Raw*>(a.in + at);
    return p + blk * BLK;
  };
  auto widen = [](const Raw& r) { return make_float4((float)r.x, (float)r.y, (float)r.z, (float)r.w); };
  float4 cur[BLK];
  Raw nxt[BLK];
  {
    const Raw* p = block_ptr(0);
#pragma unroll
    for (int q = 0; q < BLK; ++q) cur[q] = widen(p[q]);
  }
  for (long k = 0; k < total; ++k) {
    {
      const Raw* p = block_ptr(k + 1 < total ? k + 1 : k);     // last block: a harmless re-read
#pragma unroll
      for (int q = 0; q < BLK; ++q) nxt[q] = p[q];
    }
    float4* d4 = reinterpret_cast<float4*>(dst) + k * BLK;          // frames are contiguous in xhp: block k of the call
#pragma unroll
    for (int q = 0; q < BLK; ++q) {
      const float xin[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
      float yo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xi = xin[e];
        const float yi = xi + m0;
        const double dx = (double)xi, dy = (double)yi;
        // b0 dx - a0 dy = -2 (dx + (a0 / 2) dy): both products are exact in f64 (f32 x f32), scaling by 2 commutes
        // with the rounding, so u and the fused add below round exactly where the reference's sub and add do
        const double u = __fma_rn(a0h, dy, dx);
        m0 = (float)__fma_rn(-2.0, u, (double)m1);
        m1 = (float)__fma_rn(-a1, dy, dx);
        yo[e] = yi;
      }
      d4[q] = make_float4(yo[0], yo[1], yo[2], yo[3]);
    }
#pragma unroll
    for (int q = 0; q < BLK; ++q) cur[q] = widen(nxt[q]);
  }
  a.hp_mem[2 * b] = m0;
  a.hp_mem[2 * b + 1] = m1;
}
template <bool S16>
__global__ __launch_bounds__(WAVE) RN_HP_VGPR_CAP void rn_highpass_kernel(RnArgs a) { rn_highpass_body<RN_HP_BLK, S16>(a); }
template <bool S16>
__global__ __launch_bounds__(WAVE) void rn_highpass_deep_kernel(RnArgs a) { rn_highpass_body<8, S16>(a); }

// keep the last RN_HIST high-passed samples of every stream at the front of its xhp row
__global__ __launch_bounds__(256) void rn_roll_history_kernel(RnArgs a) {
  const int b = blockIdx.x;
  float* row = a.xhp + (long)b * a.xhp_stride;
  const long src = (long)a.T * RN_FRAME;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = threadIdx.x + 256 * q;
    v[q] = i < RN_HIST ? row[src + i] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = threadIdx.x + 256 * q;
    if (i < RN_HIST) row[i] = v[q];
  }
}

}  // namespace

hipError_t rn_launch_highpass(const RnArgs& a, hipStream_t s, bool deep) {
  const dim3 grid((a.B + WAVE - 1) / WAVE), block(WAVE);
  if (a.in_s16) {
    if (deep) hipLaunchKernelGGL(rn_highpass_deep_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(rn_highpass_kernel<true>, grid, block, 0, s, a);
  } else {
    if (deep) hipLaunchKernelGGL(rn_highpass_deep_kernel<false>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(rn_highpass_kernel<false>, grid, block, 0, s, a);
  }
  return hipGetLastError();
}
hipError_t rn_launch_roll_history(const RnArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(rn_roll_history_kernel, dim3(a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}


}  // namespace crispy
