// fft_lds.h -- wave-level mixed-radix Stockham FFT passes on complex data held in LDS.
// One wave, in place: every lane reads the inputs of its butterflies into registers, the
// workgroup-of-one-wave (or the caller's wave-local barrier) synchronises, every lane writes.
#pragma once
#include <hip/hip_runtime.h>

namespace crispy {
namespace fftx {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

template <int R>
__device__ __forceinline__ void butterfly(const float2 (&v)[R], float2 (&o)[R]);
template <>
__device__ __forceinline__ void butterfly<2>(const float2 (&v)[2], float2 (&o)[2]) {
  o[0] = cadd(v[0], v[1]);
  o[1] = csub(v[0], v[1]);
}
template <>
__device__ __forceinline__ void butterfly<4>(const float2 (&v)[4], float2 (&o)[4]) {
  const float2 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
  const float2 s1 = cadd(v[1], v[3]), d1 = csub(v[1], v[3]);
  o[0] = cadd(s0, s1);
  o[2] = csub(s0, s1);
  o[1] = make_float2(d0.x + d1.y, d0.y - d1.x);
  o[3] = make_float2(d0.x - d1.y, d0.y + d1.x);
}
template <>
__device__ __forceinline__ void butterfly<5>(const float2 (&v)[5], float2 (&o)[5]) {
  const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
  const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
  const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
  const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
  o[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
  const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
  const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
  const float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
  const float2 b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
  o[1] = make_float2(a1.x + b1.y, a1.y - b1.x);
  o[4] = make_float2(a1.x - b1.y, a1.y + b1.x);
  o[2] = make_float2(a2.x + b2.y, a2.y - b2.x);
  o[3] = make_float2(a2.x - b2.y, a2.y + b2.x);
}

// Wave-local barrier for LDS traffic of ONE wave inside a multi-wave workgroup: the wave runs in
// lockstep and its LDS operations complete in order; the fence keeps the compiler from moving
// LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One Stockham pass of an N-point FFT: radix R, NS = product of the earlier radices.
// tw[] holds exp(-2 pi i k / WN) with WN a multiple of N.
template <int N, int R, int NS, int WN>
__device__ __forceinline__ void pass(float2* buf, const float2* __restrict__ tw, int lane) {
  constexpr int M = N / R;
  constexpr int NBF = (M + 63) / 64;
  float2 o[NBF][R];
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int j = lane + 64 * nb;
    if (j < M) {
      const int k = j % NS;
      float2 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float2 x = buf[j + r * M];
        if (NS > 1 && r > 0) x = cmul(x, tw[k * r * (WN / (NS * R))]);
        v[r] = x;
      }
      butterfly<R>(v, o[nb]);
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int j = lane + 64 * nb;
    if (j < M) {
      const int k = j % NS;
      const int j0 = (j / NS) * NS * R + k;
#pragma unroll
      for (int r = 0; r < R; ++r) buf[j0 + r * NS] = o[nb][r];
    }
  }
  wave_lds_sync();
}

// The same pass over NF transforms at once (buffers STRIDE complex numbers apart): the NF * M butterflies are spread
// over the 64 lanes, so a radix with M = 40 or 50 butterflies no longer leaves lanes idle, and the two wave
// synchronisations of a pass are paid once per NF transforms.
template <int N, int R, int NS, int WN, int NF, int STRIDE>
__device__ __forceinline__ void pass_batched(float2* buf, const float2* __restrict__ tw, int lane) {
  constexpr int M = N / R;
  constexpr int TOT = NF * M;
  constexpr int NBF = (TOT + 63) / 64;
  float2 o[NBF][R];
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int idx = min(lane + 64 * nb, TOT - 1);      // clamped: the extra lanes redo the last butterfly, stores are predicated
    const int f = idx / M, j = idx - f * M;
    const int k = j % NS;
    const float2* b = buf + f * STRIDE;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float2 x = b[j + r * M];
      if (NS > 1 && r > 0) x = cmul(x, tw[k * r * (WN / (NS * R))]);
      v[r] = x;
    }
    butterfly<R>(v, o[nb]);
  }
  wave_lds_sync();
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int idx = lane + 64 * nb;
    if (idx < TOT) {
      const int f = idx / M, j = idx - f * M;
      const int k = j % NS;
      const int j0 = (j / NS) * NS * R + k;
      float2* b = buf + f * STRIDE;
#pragma unroll
      for (int r = 0; r < R; ++r) b[j0 + r * NS] = o[nb][r];
    }
  }
  wave_lds_sync();
}

}  // namespace fftx
}  // namespace crispy
