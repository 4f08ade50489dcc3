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

__device__ __forceinline__ int acc_row_h(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_f16_nt_kernel(GemmArgs g, const _Float16* __restrict__ Wh) {
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

  // epilogue: residual / row-table operands of eight rows requested together from clamped addresses (see
  // gemm_f32_nt_kernel), only the stores are predicated
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 32 * j + li;
      const int nc = min(n, g.N - 1);
      const float bias = g.bias ? g.bias[nc] : 0.f;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        float extra[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) extra[r] = 0.f;
        if (g.residual) {
          const float* rp = g.residual + (long)bz * g.strideR + nc;
#pragma unroll
          for (int r = 0; r < 8; ++r)
            extra[r] = rp[(long)min(m0 + wm + 32 * i + acc_row_h(r0 + r, lane), g.M - 1) * g.ldr];
        }
        if (g.rowtab) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            extra[r] += g.rowtab[(long)(min(m0 + wm + 32 * i + acc_row_h(r0 + r, lane), g.M - 1) % g.rowtab_period) * g.N + nc];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int m = m0 + wm + 32 * i + acc_row_h(r0 + r, lane);
          float v = acc[i][j][r0 + r] + bias;
          if (g.gelu) v = gelu_ggml(v);       // f16-operand GEMM: precision mode 1 only
          v += extra[r];
          if (m < g.M && n < g.N) C[(long)m * g.ldc + n] = v;
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------
// Encoder self-attention with f16 operands (same mode switch).  Same decomposition as attn_enc_kernel
// (whisper_kernels.hip): wave = 32 queries, S^T = K.Q^T per 32-key tile, online softmax in f32 per lane,
// O^T += V^T.P^T with P^T taken straight from the S^T accumulator registers.  With 16-deep MFMAs a tile is 4 + 4
// instructions instead of 32 + 32.  The contraction order of the second product is free, so k-slot (lane half lh,
// element e) of step ks is *defined* as the key held in this lane's S^T register 8 ks + e: no cross-lane exchange
// of probabilities, V is simply gathered in that key order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_enc_f16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int T, int D) {
  __shared__ float stage[4][32 * 65];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= T) return;
  const long ld = 3L * D;
  const float* base = qkv + (long)b * T * ld;
  const float* Qp = base + h * 64;
  const float* Kp = base + D + h * 64;
  const float* Vp = base + 2 * D + h * 64;

  auto load8 = [](const float* p, float scale) {      // 8 consecutive floats -> half8
    const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
    half8 r;
    r[0] = (_Float16)(a.x * scale); r[1] = (_Float16)(a.y * scale); r[2] = (_Float16)(a.z * scale); r[3] = (_Float16)(a.w * scale);
    r[4] = (_Float16)(c.x * scale); r[5] = (_Float16)(c.y * scale); r[6] = (_Float16)(c.z * scale); r[7] = (_Float16)(c.w * scale);
    return r;
  };
  // Q operand of k-step ks: Q[q = li][16 ks + 8 lh .. + 7], pre-scaled by 1/8 (exact)
  half8 qh[4];
  {
    const int q = min(q0 + li, T - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qh[ks] = load8(Qp + (long)q * ld + 16 * ks + 8 * lh, 0.125f);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  for (int k0 = 0; k0 < T; k0 += 32) {
    // ---- S^T tile: rows = keys (li), columns = queries ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const int key = min(k0 + li, T - 1);
      half8 kh[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kh[ks] = load8(Kp + (long)key * ld + 16 * ks + 8 * lh, 1.f);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks], qh[ks], s, 0, 0, 0);
    }
    // V gather issued before the softmax arithmetic: register r of this lane is key (r&3) + 8(r>>2) + 4 lh
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(k0 + acc_row_h(r, lane), T - 1);
      const float* vp = Vp + (long)key * ld;
      v0[r] = vp[li];
      v1[r] = vp[32 + li];
    }
    // ---- online softmax over this lane's 16 keys + the partner half's 16 ----
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row_h(r, lane);
      if (key >= T) s[r] = -1e30f;
      mloc = fmaxf(mloc, s[r]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
    half8 ph[2], vh0[2], vh1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(s[r] - m_new);
      psum += p;
      ph[r >> 3][r & 7] = (_Float16)p;
      vh0[r >> 3][r & 7] = (_Float16)v0[r];
      vh1[r >> 3][r & 7] = (_Float16)v1[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- O^T += V^T . P^T, k-slot (lh, e) of step ks = the key of register 8 ks + e ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh0[ks], ph[ks], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh1[ks], ph[ks], o1, 0, 0, 0);
    }
  }
  // ---- normalise, transpose through LDS, store rows of 64 floats ----
  const float inv = 1.f / l_run;
  float* st = stage[wave];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = acc_row_h(r, lane);
    st[li * 65 + d] = o0[r] * inv;
    st[li * 65 + 32 + d] = o1[r] * inv;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int idx = lane; idx < 32 * 64; idx += 64) {
    const int q = idx >> 6, d = idx & 63;
    if (q0 + q < T) out[((long)b * T + q0 + q) * D + h * 64 + d] = st[q * 65 + d];
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
hipError_t attn_encoder_f16(const float* qkv, float* out, int B, int T, int D, int heads, hipStream_t s) {
  hipLaunchKernelGGL(attn_enc_f16_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, s, qkv, out, T, D);
  return hipGetLastError();
}
hipError_t convert_f32_to_f16(const float* src, void* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                     reinterpret_cast<_Float16*>(dst), n);
  return hipGetLastError();
}

}  // namespace crispy
