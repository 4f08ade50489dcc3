// whisper_gemm_f16.hip -- f16-operand GEMM for the Whisper encoder on MI355X (gfx950), opt-in
// (crispy_asr_set_precision(h, 1)).
//
// whisper.cpp runs its matrix products with f16 weights and the f32 activations rounded to f16, accumulating in
// f32 (ggml mul_mat) [UPSTREAM-RECALL]; this kernel has the same numerics: W is stored as f16 (converted once when
// the mode is switched on), A is f32 in HBM and rounded to f16 while it is staged into LDS, products and sums are
// f32 on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense peak, 16x the f32-operand rate the default path uses).
//
//   C[M,N] = f16(A[M,K]) . Wh[N,K]^T (+bias) (GELU) (+residual | +row-periodic table), f32 out
//   128x128x32 tiles, 4 waves x (2x2) MFMA 32x32 tiles, LDS double buffer with 40-half rows (conflict-free b128).
// A may be a strided view (lda < K) exactly as in gemm_f32_nt_kernel.
#include "asr_common.h"

namespace crispy {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int HB_M = 128, HB_N = 128, HB_K = 32, HB_LD = 40;

__device__ __forceinline__ float gelu_erf_h(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ int acc_row_h(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(256) void gemm_f16_nt_kernel(GemmArgs g, const _Float16* __restrict__ Wh) {
  __shared__ __attribute__((aligned(16))) _Float16 As[2][HB_M * HB_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Ws[2][HB_N * HB_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bz = blockIdx.z;
  const float* __restrict__ A = g.A + (long)bz * g.strideA;
  float* __restrict__ C = g.C + (long)bz * g.strideC;
  const int m0 = blockIdx.y * HB_M, n0 = blockIdx.x * HB_N;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging: A as float4 -> half4 (thread: row (tid >> 3) + 32 h, k offset 4 (tid & 7)); W as 8 halfs (row
  // (tid >> 2) + 64 h, k offset 8 (tid & 3))
  const int ar = tid >> 3, ak = (tid & 7) * 4;
  const int wr = tid >> 2, wk = (tid & 3) * 8;
  float4 ra[4];
  uint4 rw[2];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int m = m0 + ar + 32 * h;
      ra[h] = (m < g.M) ? *reinterpret_cast<const float4*>(A + (long)m * g.lda + k0 + ak) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + wr + 64 * h;
      rw[h] = (n < g.N) ? *reinterpret_cast<const uint4*>(Wh + (long)n * g.ldw + k0 + wk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      half4 v;
      v[0] = (_Float16)ra[h].x; v[1] = (_Float16)ra[h].y; v[2] = (_Float16)ra[h].z; v[3] = (_Float16)ra[h].w;
      *reinterpret_cast<half4*>(&As[buf][(ar + 32 * h) * HB_LD + ak]) = v;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<uint4*>(&Ws[buf][(wr + 64 * h) * HB_LD + wk]) = rw[h];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / HB_K;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kb = 0; kb < nk; ++kb) {
    const int buf = kb & 1;
    if (kb + 1 < nk) load_tiles((kb + 1) * HB_K);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // lane supplies 8 consecutive k of its row / column: k = 16 ks + 8 lh .. + 7
      half8 a[2], w[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const half8*>(&As[buf][(wm + 32 * i + li) * HB_LD + 16 * ks + 8 * lh]);
        w[i] = *reinterpret_cast<const half8*>(&Ws[buf][(wn + 32 * i + li) * HB_LD + 16 * ks + 8 * lh]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], w[j], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 32 * j + li;
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + acc_row_h(r, lane);
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bias;
        if (g.gelu) v = gelu_erf_h(v);
        if (g.residual) v += g.residual[(long)bz * g.strideR + (long)m * g.ldr + n];
        if (g.rowtab) v += g.rowtab[(long)(m % g.rowtab_period) * g.N + n];
        C[(long)m * g.ldc + n] = v;
      }
    }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (_Float16)src[i];
}

}  // namespace

// K must be a multiple of 32 and A rows 16-byte aligned (the caller checks); Wh: [N][ldw] halfs
hipError_t gemm_f16_nt(const GemmArgs& g, const void* Wh, int batch, hipStream_t s) {
  dim3 grid((g.N + HB_N - 1) / HB_N, (g.M + HB_M - 1) / HB_M, batch);
  hipLaunchKernelGGL(gemm_f16_nt_kernel, grid, dim3(256), 0, s, g, reinterpret_cast<const _Float16*>(Wh));
  return hipGetLastError();
}
hipError_t convert_f32_to_f16(const float* src, void* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                     reinterpret_cast<_Float16*>(dst), n);
  return hipGetLastError();
}

}  // namespace crispy
