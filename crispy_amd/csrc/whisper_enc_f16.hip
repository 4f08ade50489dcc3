// whisper_enc_f16.hip -- the Whisper encoder at the reference's precision on MI355X (gfx950): precision mode 1.
//
// whisper.cpp computes every matrix product with f16 operands and f32 accumulation (ggml mul_mat: f16 weights, the f32
// activations converted to f16 on the way in) [UPSTREAM-RECALL].  An activation that ONLY feeds a matrix product can
// therefore be stored already rounded -- same numbers, half the bytes: the LayerNorm outputs, q | k | v, the attention
// output and the GELU'd MLP hidden layer live in HBM as f16; the residual stream, which is added to, stays f32.
//
//   layernorm_h_kernel     f32 row -> f16 row (statistics in f32)
//   gemm_hh_kernel<EPI>    C = A[M,K] (f16) . W[N,K]^T (f16), f32 accumulation on v_mfma_f32_32x32x16_f16,
//                          128 x 128 x 32 tiles, 4 waves x (2 x 2) MFMA tiles, double-buffered LDS (40-half rows).
//                            EPI_F16   +bias (, GELU) -> f16 row-major                     (q | k, fc1)
//                            EPI_RES   +bias +residual -> f32 row-major, in place          (attention out-proj, fc2)
//                            EPI_VT    +bias -> f16 V^T[clip][head][64][ENC_TP]           (v, transposed for attention)
//   attn_enc_h_kernel      flash-style attention on f16 q | k and V^T: wave = 32 queries, S^T = K.Q^T per 32-key tile,
//                          softmax statistics per lane in f32 (one v_permlane32_swap per tile joins the two key halves),
//                          P rounded to f16 straight from the S^T accumulator registers into the B operand of
//                          O^T += V^T.P^T; the running maximum is only raised when a score exceeds it by 2^6 (a stale
//                          maximum scales every probability of a row by the same power of two, which f16 rounding does
//                          not see), so the 32 accumulator rescales per tile are rare.
#include "api_util.h"
#include <type_traits>
#include "asr_common.h"

#include <cstdlib>

namespace crispy {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int HH_M = 128, HH_N = 128, HH_K = 32, HH_LD = 40;

// GELU: ggml's form (asr_common.h: gelu_ggml).  (Rounds 1 - 2 used the exact erf form here -- Abramowitz & Stegun 7.1.26,
// on packed f32 instructions: 97 cycles per value and lane, 12.4 k of the 23.4 k cycles of an fc1 epilogue.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// row index of accumulator register r for this lane (32 x 32 MFMA C / D layout)
__device__ __forceinline__ int acc_row_e(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ half4 to_half4(float a, float b, float c, float d) {
  const half2v lo = __builtin_convertvector(float2v{a, b}, half2v), hi = __builtin_convertvector(float2v{c, d}, half2v);
  return half4{lo[0], lo[1], hi[0], hi[1]};
}

template <int PER>
__global__ __launch_bounds__(256) void layernorm_h_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, _Float16* __restrict__ y,
                                                          long rows) {
  constexpr int D = 64 * PER;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * D;
  float v[PER], gm[PER], bt[PER];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) v[q] = xr[lane + 64 * q];
#pragma unroll
  for (int q = 0; q < PER; ++q) { gm[q] = gamma[lane + 64 * q]; bt[q] = beta[lane + 64 * q]; }
#pragma unroll
  for (int q = 0; q < PER; ++q) s += v[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) { const float d = v[q] - mean; s2 = fmaf(d, d, s2); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
  const float rstd = 1.f / sqrtf(s2 / (float)D + 1e-5f);
  _Float16* yr = y + row * D;
#pragma unroll
  for (int q = 0; q < PER; ++q) yr[lane + 64 * q] = (_Float16)((v[q] - mean) * rstd * gm[q] + bt[q]);
}

enum { EPI_F16 = 0, EPI_RES = 1, EPI_VT = 2, EPI_TAB = 3, EPI_KVH = 4, EPI_F32 = 5 };

// one k-block (32) of a wave's 64 x 64 output tile from the staged operand tiles
template <int EPI>
__device__ __forceinline__ void hh_compute(const _Float16* Asb, const _Float16* Wsb, f32x16 (&acc)[2][2], int wm, int wn,
                                           int li, int lh) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    half8 a[2], w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const half8*>(&Asb[(wm + 32 * i + li) * HH_LD + 16 * ks + 8 * lh]);
      w[i] = *reinterpret_cast<const half8*>(&Wsb[(wn + 32 * i + li) * HH_LD + 16 * ks + 8 * lh]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // EPI_F16 wants a lane to own a ROW of C (four consecutive columns per register group -> 8-byte stores):
        // D = W.A^T puts m in the lane and n in the registers.  The other two want a lane to own a COLUMN.
        if (EPI == EPI_F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], w[j], acc[i][j], 0, 0, 0);
      }
  }
}

// The epilogues, shared by the register-staged and the LDS-direct main loops.
template <int EPI>
__device__ __forceinline__ void hh_epilogue(const HGemmArgs& g, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int lane,
                                            int bz) {
  const int li = lane & 31, lh = lane >> 5;
  if (EPI == EPI_F16) {
    _Float16* __restrict__ C = reinterpret_cast<_Float16*>(g.C) + (long)bz * g.strideC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm + 32 * i + li;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = min(n0 + wn + 32 * j + 8 * q + 4 * lh, g.N - 4);
          bq[q] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn + 32 * j + 8 * q + 4 * lh;
          float v0 = acc[i][j][4 * q] + bq[q].x, v1 = acc[i][j][4 * q + 1] + bq[q].y;
          float v2 = acc[i][j][4 * q + 2] + bq[q].z, v3 = acc[i][j][4 * q + 3] + bq[q].w;
          if (g.gelu) { v0 = gelu_ggml(v0); v1 = gelu_ggml(v1); v2 = gelu_ggml(v2); v3 = gelu_ggml(v3); }
          if (m < g.M && n < g.N) *reinterpret_cast<half4*>(C + (long)m * g.ldc + n) = to_half4(v0, v1, v2, v3);
        }
      }
    }
  } else if (EPI == EPI_RES) {
    // wave-uniform tile base pointers + 32-bit lane offsets: 64-bit per-lane addresses for C and the residual pushed
    // this epilogue into scratch at four waves per SIMD
    float* __restrict__ Ct = reinterpret_cast<float*>(g.C) + (long)bz * g.strideC + (long)(m0 + wm) * g.ldc + (n0 + wn);
    const float* __restrict__ Rt = g.residual + (long)bz * g.strideC + (long)(m0 + wm) * g.ldr + (n0 + wn);
    const int ldc = (int)g.ldc, ldr = (int)g.ldr;
    const int mrem = g.M - (m0 + wm);                 // rows of this wave's 64-row band that exist
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = 32 * j + li;
        const bool nok = n0 + wn + nl < g.N;
        const float bias = g.bias ? g.bias[min(n0 + wn + nl, g.N - 1)] : 0.f;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
          float extra[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)     // residual operands of four rows requested together (clamped rows)
            extra[r] = Rt[min(32 * i + acc_row_e(r0 + r, lane), mrem - 1) * ldr + (nok ? nl : 0)];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ml = 32 * i + acc_row_e(r0 + r, lane);
            if (ml < mrem && nok) Ct[ml * ldc + nl] = acc[i][j][r0 + r] + bias + extra[r];
          }
        }
      }
  } else if (EPI == EPI_TAB) {
    // conv2: f32 out = GELU(acc + bias) + positional row (m % period); same addressing as EPI_RES
    float* __restrict__ Ct = reinterpret_cast<float*>(g.C) + (long)bz * g.strideC + (long)(m0 + wm) * g.ldc + (n0 + wn);
    const float* __restrict__ Tt = g.rowtab + (n0 + wn);
    const int ldc = (int)g.ldc;
    const int mrem = g.M - (m0 + wm);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = 32 * j + li;
        const bool nok = n0 + wn + nl < g.N;
        const float bias = g.bias ? g.bias[min(n0 + wn + nl, g.N - 1)] : 0.f;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
          float extra[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mg = min(m0 + wm + 32 * i + acc_row_e(r0 + r, lane), g.M - 1) % g.rowtab_period;
            extra[r] = Tt[mg * g.N + (nok ? nl : 0)];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ml = 32 * i + acc_row_e(r0 + r, lane);
            if (ml < mrem && nok) Ct[ml * ldc + nl] = gelu_ggml(acc[i][j][r0 + r] + bias) + extra[r];
          }
        }
      }
  } else {   // EPI_VT: column n = (head, dim), row m = (clip, t) -> Vt[((clip * heads + head) * 64 + dim) * ENC_TP + t]
    _Float16* __restrict__ C = reinterpret_cast<_Float16*>(g.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        const int nc = min(n, g.N - 1);
        const float bias = g.bias ? g.bias[nc] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wm + 32 * i + 8 * q + 4 * lh;       // four consecutive rows; T % 4 == 0, so one clip
          const int clip = m / g.vt_T, t = m - clip * g.vt_T;
          const half4 v = to_half4(acc[i][j][4 * q] + bias, acc[i][j][4 * q + 1] + bias, acc[i][j][4 * q + 2] + bias,
                                   acc[i][j][4 * q + 3] + bias);
          if (m < g.M && n < g.N) *reinterpret_cast<half4*>(C + ((long)clip * g.N + nc) * ENC_TP + t) = v;
        }
      }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void gemm_hh_kernel(HGemmArgs g) {
  __shared__ __attribute__((aligned(16))) _Float16 As[2][HH_M * HH_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Ws[2][HH_N * HH_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Workgroups are dealt to the 8 XCDs round robin.  All column tiles of one row tile go to the same XCD (its L2 then
  // serves the A tile to all of them, instead of eight L2s fetching it once each): linear id -> (xcd, slot) ->
  // row tile = 8 (slot / n_tiles) + xcd, column tile = slot % n_tiles.  Row tiles past the matrix return at once.
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_swizzle) {
    const int nt = gridDim.x;
    const int lin = blockIdx.y * nt + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    by = 8 * (slot / nt) + xcd;
    bx = slot % nt;
    if (by * HH_M >= g.M) return;
  }
  const int bz = blockIdx.z;
  const _Float16* __restrict__ A = g.A + (long)bz * g.strideA;
  const _Float16* __restrict__ W = g.W;
  const int m0 = by * HH_M, n0 = bx * HH_N;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging: 8 halfs (16 bytes) per thread and tile half: row (tid >> 2) + 64 h, k offset 8 (tid & 3).  Plain named
  // registers and unconditional code: with the loads behind `if (kb + 1 < nk)` in a lambda over arrays the compiler
  // kept the staging registers in SCRATCH (load - wait - scratch store, scratch load - LDS store: no prefetch at all).
  const int sr = tid >> 2, sk = (tid & 3) * 8;
  const int M = g.M, N = g.N;
  const long lda = g.lda, ldw = g.ldw;
  const _Float16* pa0 = A + (long)min(m0 + sr, M - 1) * lda + sk;        // clamped rows: their results are not stored
  const _Float16* pa1 = A + (long)min(m0 + sr + 64, M - 1) * lda + sk;
  const _Float16* pw0 = W + (long)min(n0 + sr, N - 1) * ldw + sk;
  const _Float16* pw1 = W + (long)min(n0 + sr + 64, N - 1) * ldw + sk;
  const int so0 = sr * HH_LD + sk, so1 = (sr + 64) * HH_LD + sk;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One k-block of operands in flight per thread, four waves per SIMD.  (Two blocks in flight -- a second register
  // set, 16 more VGPRs -- drop the kernel to three / two waves per SIMD and measured 18 % SLOWER: what hides the L2 /
  // HBM round trip here is the number of resident waves.)
  const int nk = g.K / HH_K;
  uint4 ra0 = *reinterpret_cast<const uint4*>(pa0), ra1 = *reinterpret_cast<const uint4*>(pa1);
  uint4 rw0 = *reinterpret_cast<const uint4*>(pw0), rw1 = *reinterpret_cast<const uint4*>(pw1);
  *reinterpret_cast<uint4*>(&As[0][so0]) = ra0;
  *reinterpret_cast<uint4*>(&As[0][so1]) = ra1;
  *reinterpret_cast<uint4*>(&Ws[0][so0]) = rw0;
  *reinterpret_cast<uint4*>(&Ws[0][so1]) = rw1;
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kb = 0; kb < nk; ++kb) {
    const int buf = kb & 1;
    const int kn = min(kb + 1, nk - 1) * HH_K;      // the last trip re-requests its own block (never used)
    ra0 = *reinterpret_cast<const uint4*>(pa0 + kn);
    ra1 = *reinterpret_cast<const uint4*>(pa1 + kn);
    rw0 = *reinterpret_cast<const uint4*>(pw0 + kn);
    rw1 = *reinterpret_cast<const uint4*>(pw1 + kn);
    // pin the requests here: left alone the scheduler sinks them below the MFMAs, right in front of the LDS stores
    // that consume them (load - wait - store: the whole memory latency exposed once per k-block)
    __builtin_amdgcn_sched_barrier(0);
    hh_compute<EPI>(As[buf], Ws[buf], acc, wm, wn, li, lh);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<uint4*>(&As[buf ^ 1][so0]) = ra0;
    *reinterpret_cast<uint4*>(&As[buf ^ 1][so1]) = ra1;
    *reinterpret_cast<uint4*>(&Ws[buf ^ 1][so0]) = rw0;
    *reinterpret_cast<uint4*>(&Ws[buf ^ 1][so1]) = rw1;
    __syncthreads();
  }

  hh_epilogue<EPI>(g, acc, m0, n0, wm, wn, lane, bz);
}

// ---------------------------------------------------------------------------------------------
// Epilogue through LDS (LDS-direct kernel).  With a lane owning a row or a column of the MFMA tile, every store
// instruction of the direct epilogues above touches 32 different rows with 8 or 16 bytes each, 16 to 64 instructions
// per wave -- store-ISSUE bound: for the K = 384 GEMMs the store tail was longer than the main loop.  Here a wave
// puts its 64 x 64 tile through a padded f32 image in LDS, 32 rows at a time, and reads it back row-contiguous:
// every global instruction then moves whole 128 / 256-byte rows (8 or 4 rows per instruction, 16 bytes per lane).
//   T: this wave's 32 x 68 floats (row-major outputs) or 64 x 36 floats (V^T) of LDS.
// ---------------------------------------------------------------------------------------------
constexpr int EP_LD = 68, EP_VLD = 36;
constexpr int EP_IMAGE_BYTES = 64 * EP_VLD * 4;      // 9216: the larger of the two image shapes (32 x 68 and 64 x 36 floats)
__device__ __forceinline__ void ep_wave_sync() {     // this wave's LDS traffic has landed (single-wave hand-off)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The three pieces of one 32-row pass.  rb = first row of the pass inside the wave's rows (m0 + wm is the wave's origin).
// bias[j]: this lane's column 32 j + li of the wave's 64, requested by the caller above the k loop.
//   ep_write     registers -> image: lane = column 32 j + li, register r = row 8 (r >> 2) + 4 lh + (r & 3)
template <int EPI>
__device__ __forceinline__ void ep_write(const HGemmArgs& g, const f32x16 (&acc)[2], float* T, int lane, const float (&bias)[2]) {
  const int li = lane & 31, lh = lane >> 5;
  if (EPI == EPI_VT) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(T + (32 * j + li) * EP_VLD + 8 * q + 4 * lh) =
            make_float4(acc[j][4 * q] + bias[j], acc[j][4 * q + 1] + bias[j], acc[j][4 * q + 2] + bias[j], acc[j][4 * q + 3] + bias[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {acc[j][r] + bias[j], acc[j][r + 1] + bias[j]};
        if ((EPI == EPI_F16 && g.gelu) || EPI == EPI_TAB) v = f32x2{gelu_ggml(v.x), gelu_ggml(v.y)};
        T[acc_row_e(r, lane) * EP_LD + 32 * j + li] = v.x;
        T[acc_row_e(r + 1, lane) * EP_LD + 32 * j + li] = v.y;
      }
  }
}
//   ep_prefetch  EPI_RES / EPI_TAB: the residual / positional operands of the pass, requested together
template <int EPI>
__device__ __forceinline__ void ep_prefetch(const HGemmArgs& g, float4 (&ex)[8], int m0, int n0, int wm, int wn, int rb, int lane, int bz) {
  if (EPI != EPI_RES && EPI != EPI_TAB) return;
  const int mrem = g.M - (m0 + wm), nrem = g.N - (n0 + wn);
  const int c4 = (lane & 15) * 4;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = (lane >> 4) + 4 * p;
    const int rc = min(rb + row, mrem - 1), cc = min(c4, nrem - 4);
    if (EPI == EPI_RES)
      ex[p] = *reinterpret_cast<const float4*>(g.residual + (long)bz * g.strideC + (long)(m0 + wm + rc) * g.ldr + (n0 + wn) + cc);
    else
      ex[p] = *reinterpret_cast<const float4*>(g.rowtab + (long)((m0 + wm + rc) % g.rowtab_period) * g.N + (n0 + wn) + cc);
  }
}
//   ep_store     image -> global, row-contiguous
template <int EPI>
__device__ __forceinline__ void ep_store(const HGemmArgs& g, const float* T, const float4 (&ex)[8], int m0, int n0, int wm, int wn,
                                         int rb, int lane, int bz) {
  const int mrem = g.M - (m0 + wm), nrem = g.N - (n0 + wn);
  if (EPI == EPI_KVH) {
    // cross K | V, head-major: this wave's 64 columns are exactly one head of K or of V (wn and the tile origin are
    // multiples of 64), a row is one frame of one clip -> 128 contiguous bytes at [clip][K|V][head][frame][64]
    _Float16* __restrict__ C = reinterpret_cast<_Float16*>(g.C);
    const int c8 = (lane & 7) * 8;
    const int ncol = n0 + wn, kv = ncol / g.kv_width, head = (ncol - kv * g.kv_width) >> 6, heads = g.kv_width >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = (lane >> 3) + 8 * p;
      const float4 x0 = *reinterpret_cast<const float4*>(T + row * EP_LD + c8);
      const float4 x1 = *reinterpret_cast<const float4*>(T + row * EP_LD + c8 + 4);
      const half4 h0 = to_half4(x0.x, x0.y, x0.z, x0.w), h1 = to_half4(x1.x, x1.y, x1.z, x1.w);
      const half8 hv = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const int m = m0 + wm + rb + row;
      const int clip = m / g.vt_T, t = m - clip * g.vt_T;
      if (m < g.M && ncol < g.N)
        *reinterpret_cast<half8*>(C + ((((long)clip * 2 + kv) * heads + head) * g.vt_T + t) * 64 + c8) = hv;
    }
  } else if (EPI == EPI_F16) {
    _Float16* __restrict__ C = reinterpret_cast<_Float16*>(g.C) + (long)bz * g.strideC + (long)(m0 + wm + rb) * g.ldc + (n0 + wn);
    const int c8 = (lane & 7) * 8;                            // 8 lanes per row, 8 columns (16 bytes of f16) each
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = (lane >> 3) + 8 * p;
      const float4 x0 = *reinterpret_cast<const float4*>(T + row * EP_LD + c8);
      const float4 x1 = *reinterpret_cast<const float4*>(T + row * EP_LD + c8 + 4);
      const half4 h0 = to_half4(x0.x, x0.y, x0.z, x0.w), h1 = to_half4(x1.x, x1.y, x1.z, x1.w);
      const half8 hv = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      if (rb + row < mrem && c8 < nrem) *reinterpret_cast<half8*>(C + (long)row * g.ldc + c8) = hv;
    }
  } else if (EPI == EPI_F32) {
    float* __restrict__ C = reinterpret_cast<float*>(g.C) + (long)bz * g.strideC + (long)(m0 + wm + rb) * g.ldc + (n0 + wn);
    const int c2 = (lane & 31) * 2;                           // 32 lanes per row, 2 columns each: rows are 8-byte aligned only
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int row = (lane >> 5) + 2 * p;
      const float2 x = *reinterpret_cast<const float2*>(T + row * EP_LD + c2);
      if (g.c_group_rows > 0) {                               // rows in groups: (group, row in group) from the global row
        const int m = m0 + wm + rb + row;
        const int grp = m / g.c_group_rows, r = m - grp * g.c_group_rows;
        if (rb + row < mrem && c2 < nrem && r < g.c_group_valid)
          *reinterpret_cast<float2*>(reinterpret_cast<float*>(g.C) + (long)grp * g.c_group_stride + (long)r * g.ldc + (n0 + wn) + c2) = x;
      } else if (rb + row < mrem && c2 < nrem) {
        *reinterpret_cast<float2*>(C + (long)row * g.ldc + c2) = x;
      }
    }
  } else if (EPI == EPI_RES || EPI == EPI_TAB) {
    float* __restrict__ C = reinterpret_cast<float*>(g.C) + (long)bz * g.strideC + (long)(m0 + wm + rb) * g.ldc + (n0 + wn);
    const int c4 = (lane & 15) * 4;                           // 16 lanes per row, 4 columns (16 bytes of f32) each
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = (lane >> 4) + 4 * p;
      const float4 x = *reinterpret_cast<const float4*>(T + row * EP_LD + c4);
      if (rb + row < mrem && c4 < nrem)
        *reinterpret_cast<float4*>(C + (long)row * g.ldc + c4) = make_float4(x.x + ex[p].x, x.y + ex[p].y, x.z + ex[p].z, x.w + ex[p].w);
    }
  } else {   // EPI_VT: image row = column n (head, dim), 32 consecutive time steps of one clip (T % 32 == ... see below)
    _Float16* __restrict__ C = reinterpret_cast<_Float16*>(g.C);
    const int c4 = (lane & 7) * 4;                            // 8 lanes per row, 4 time steps (8 bytes of f16) each
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int nl = (lane >> 3) + 8 * p;                     // column of this wave's 64
      const float4 x = *reinterpret_cast<const float4*>(T + nl * EP_VLD + c4);
      const int m = m0 + wm + rb + c4;                        // four consecutive rows; vt_T % 4 == 0, so one clip
      const int clip = m / g.vt_T, t = m - clip * g.vt_T;
      if (m < g.M && nl < nrem)
        *reinterpret_cast<half4*>(C + ((long)clip * g.N + (n0 + wn + nl)) * ENC_TP + t) = to_half4(x.x, x.y, x.z, x.w);
    }
  }
}
// 64 x 64 wave tile, one image: write, hand-off, store, hand-off -- twice
template <int EPI>
__device__ __forceinline__ void hd_epilogue(const HGemmArgs& g, f32x16 (&acc)[2][2], float* T, int m0, int n0, int wm, int wn,
                                            int lane, int bz, const float (&bias)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float4 ex[8];
    ep_write<EPI>(g, acc[i], T, lane, bias);
    ep_wave_sync();
    ep_prefetch<EPI>(g, ex, m0, n0, wm, wn, 32 * i, lane, bz);       // (ahead of the image write it costs 12 - 18 registers: 128 is the budget here)
    ep_store<EPI>(g, T, ex, m0, n0, wm, wn, 32 * i, lane, bz);
    ep_wave_sync();                                  // the image is read before the second half overwrites it
  }
}
// 128 x 64 wave tile, two images: the image of pass p + 1 is written BEFORE the hand-off of pass p, so a pass is one
// LDS round trip instead of two (measured on the one-image form: 1 750 cycles from the first image write to the
// hand-off and 1 790 from there to the last store -- each mostly queueing behind the other workgroup's k loop), and
// the residual / positional operands of a pass are requested a whole pass ahead.
template <int EPI, int MI>
__device__ __forceinline__ void hd2_epilogue(const HGemmArgs& g, f32x16 (&acc)[MI][2], float* T0, float* T1, int m0, int n0, int wm,
                                             int wn, int lane, int bz, const float (&bias)[2]) {
  float4 ex[2][8];
  ep_prefetch<EPI>(g, ex[0], m0, n0, wm, wn, 0, lane, bz);
  ep_write<EPI>(g, acc[0], T0, lane, bias);
#pragma unroll
  for (int p = 0; p < MI; ++p) {
    if (p + 1 < MI) {
      ep_prefetch<EPI>(g, ex[(p + 1) & 1], m0, n0, wm, wn, 32 * (p + 1), lane, bz);
      ep_write<EPI>(g, acc[p + 1], (p & 1) ? T0 : T1, lane, bias);     // its previous reader (pass p - 1) has issued its stores
    }
    ep_wave_sync();
    ep_store<EPI>(g, (p & 1) ? T1 : T0, ex[p & 1], m0, n0, wm, wn, 32 * p, lane, bz);
  }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM with LDS-direct operand loads and a three-stage pipeline (the default main loop).
//
// The register-staged loop above has one k-block of operands in flight per wave, issued at the top of the trip that
// stores it to LDS at its end: 256 cycles of MFMAs (~1000 with the other three waves of the SIMD) against an L2 / HBM
// round trip of several thousand under load -- matrix pipe 21 % busy.  Here global_load_lds_dwordx4 writes the operand
// tiles straight into LDS (no staging registers, no ds_write), so a third LDS stage costs no VGPRs and the requests of
// k-block kb + 2 are in flight while kb is multiplied:
//   trip kb:  s_waitcnt vmcnt(4 | 0)   this wave's four requests of stage kb have landed (kb + 1 may be pending)
//             s_barrier                 everyone's have; everyone has finished reading stage kb - 1
//             request stage kb + 2      into the buffer stage kb - 1 occupied
//             8 ds_read_b128 + 8 MFMAs from stage kb
// LDS tile layout: 128 rows x 4 chunks of 16 bytes, unpadded (a wave-wide request writes 1 KB contiguously: lane L at
// base + 16 L), chunk c of row r stored at slot c ^ ((r >> 2) & 3).  A ds_read_b128 is served in four groups of 16
// NON-contiguous lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) against 64 banks = a 256-byte bank
// row = four of these 64-byte tile rows: the lanes of a group sit on rows with r mod 4 = 0..3 four times over, and the four
// rows of one residue have (r >> 2) & 3 all different ({0,3,1,2} / {1,2,0,3}), so the group covers the 16 slots of the
// bank row exactly once.  (The first version XORed with (r >> 1) & 3, thought out for 8 consecutive lanes over 32
// banks: measured SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE, every operand read two-way conflicted.)
// The lane that fills slot (r, cs) simply requests chunk cs ^ ((r >> 2) & 3).
// The operand reads are inline asm on purpose: the compiler orders every LDS read it can see behind the most recent
// LDS-DMA request (`s_waitcnt vmcnt(0)`), which would serialise the stages; the counters are kept by hand instead.
// ---------------------------------------------------------------------------------------------
constexpr int HD_STAGES = 3, HD_TILE_BYTES = HH_M * HH_K * 2;     // 8 KB per operand tile
// where k-block k0 of A starts inside a row (HGemmArgs::k_seg: K in segments that live at their own offsets)
__device__ __forceinline__ long a_seg_k(const HGemmArgs& g, int k0) {
  if (g.k_seg <= 0) return k0;
  const int seg = k0 / g.k_seg;
  return g.a_seg_off[seg] + (k0 - seg * g.k_seg);
}
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_hd_kernel(HGemmArgs g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[HD_STAGES * 2 * HD_TILE_BYTES];   // [stage][A | W]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_swizzle) {
    const int nt = gridDim.x;
    const int lin = blockIdx.y * nt + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    by = 8 * (slot / nt) + xcd;
    bx = slot % nt;
    if (by * HH_M >= g.M) return;
  }
  const int bz = blockIdx.z;
  const _Float16* __restrict__ A = g.A + (long)bz * g.strideA;
  const _Float16* __restrict__ W = g.W;
  const int m0 = by * HH_M, n0 = bx * HH_N;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;

  // ---- requests: wave w fills rows [32 w, 32 w + 32) of both tiles, two 1 KB wave-requests each ----
  const _Float16* ga[2];
  const _Float16* gw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = 32 * wave + 16 * u + (lane >> 2);             // tile row this lane fills
    const int c = (lane & 3) ^ ((r >> 2) & 3);                  // global chunk that belongs into slot (r, lane & 3)
    ga[u] = A + (long)min(m0 + r, g.M - 1) * g.lda + 8 * c;     // clamped rows: their results are not stored
    gw[u] = W + (long)min(n0 + r, g.N - 1) * g.ldw + 8 * c;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto request = [&](int stage, int k0) {
    unsigned char* base = smem + stage * (2 * HD_TILE_BYTES) + wave * 2048;
    const long ka = a_seg_k(g, k0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      __builtin_amdgcn_global_load_lds(ga[u] + ka, (lds_ptr)(base + 1024 * u), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gw[u] + k0, (lds_ptr)(base + HD_TILE_BYTES + 1024 * u), 16, 0, 0);
    }
  };

  // ---- operand read addresses (bytes inside a stage): row R = w? + 32 i + li, chunk 2 ks + lh, slot = chunk ^ swz ----
  const int swz = (li >> 2) & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const unsigned a_ks0 = lds0 + (wm + li) * 64 + ((lh ^ swz) << 4);
  const unsigned a_ks1 = lds0 + (wm + li) * 64 + (((2 + lh) ^ swz) << 4);
  const unsigned w_ks0 = lds0 + HD_TILE_BYTES + (wn + li) * 64 + ((lh ^ swz) << 4);
  const unsigned w_ks1 = lds0 + HD_TILE_BYTES + (wn + li) * 64 + (((2 + lh) ^ swz) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / HH_K;
  request(0, 0);
  if (nk > 1) request(1, HH_K);
  float bias[2];                             // epilogue operand, requested here (behind the first operand requests)
#pragma unroll
  for (int j = 0; j < 2; ++j) bias[j] = g.bias ? g.bias[min(n0 + wn + 32 * j + li, g.N - 1)] : 0.f;
  int stage = 0;
  for (int kb = 0; kb < nk; ++kb) {
    if (kb + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4): the four requests of stage kb + 1 may be pending
    else __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (kb + 2 < nk) {
      const int s2 = stage + 2 >= HD_STAGES ? stage + 2 - HD_STAGES : stage + 2;
      request(s2, (kb + 2) * HH_K);
    }
    const unsigned so = (unsigned)stage * (2 * HD_TILE_BYTES);
    half8 a0[2], a1[2], w0[2], w1[2];          // [ks]
    asm volatile("ds_read_b128 %0, %1" : "=v"(a0[0]) : "v"(a_ks0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(w0[0]) : "v"(w_ks0 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1[0]) : "v"(a_ks0 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1[0]) : "v"(w_ks0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(a0[1]) : "v"(a_ks1 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(w0[1]) : "v"(w_ks1 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1[1]) : "v"(a_ks1 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1[1]) : "v"(w_ks1 + so));
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0[0]), "+v"(w0[0]), "+v"(a1[0]), "+v"(w1[0]));
#define HD_MFMA(i_, j_, av, wv) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc[i_][j_], 0, 0, 0);
    HD_MFMA(0, 0, a0[0], w0[0]) HD_MFMA(0, 1, a0[0], w1[0]) HD_MFMA(1, 0, a1[0], w0[0]) HD_MFMA(1, 1, a1[0], w1[0])
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[1]), "+v"(w0[1]), "+v"(a1[1]), "+v"(w1[1]));
    HD_MFMA(0, 0, a0[1], w0[1]) HD_MFMA(0, 1, a0[1], w1[1]) HD_MFMA(1, 0, a1[1], w0[1]) HD_MFMA(1, 1, a1[1], w1[1])
#undef HD_MFMA
    stage = stage + 1 == HD_STAGES ? 0 : stage + 1;
  }
  __builtin_amdgcn_s_barrier();          // every wave has read its last operands: the stages become epilogue images
  hd_epilogue<EPI>(g, acc, reinterpret_cast<float*>(smem + wave * 12288), m0, n0, wm, wn, lane, bz, bias);
}

// ---------------------------------------------------------------------------------------------
// Encoder self-attention, head dim 64, on f16 q | k (row-major [B * T][2 D]) and V^T ([B][heads][64][ENC_TP]).
// grid (ceil(T / 128), heads, B), 4 waves, wave = 32 queries; out f16 [B * T][D].
// ---------------------------------------------------------------------------------------------
constexpr int AT_KLD = 72;    // halfs per K row in LDS (64 + 8): 16-lane b128 reads spread over all banks
constexpr int AT_VLD = 36;    // halfs per V^T row in LDS (32 + 4): b64 reads of 16 rows hit 16 distinct bank pairs
// NORM16 (precision mode 2): ggml's order inside the attention -- soft-max in full, normalised, THEN rounded to f16 in
// front of P.V [UPSTREAM-RECALL] -- which needs the row's maximum and sum before the first probability is rounded: a
// statistics pass over K (scores, running maximum, running sum) in front of the pass that multiplies.
template <bool NORM16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void attn_enc_h_kernel(const _Float16* __restrict__ qk, const _Float16* __restrict__ vt,
                                                         _Float16* __restrict__ out, int T, int D, int heads, int n_clips) {
  // K and V^T tiles of 32 keys are staged once per workgroup (the four waves work on the same clip and head) and
  // double buffered: the requests for tile i + 1 are in flight while tile i is computed, one barrier per tile.
  // ONE LDS object: [K stage 0 | K stage 1 | V^T stage 0 | V^T stage 1] = 4 x 4608 bytes.  The epilogue reuses all of it as
  // four per-wave [32][AT_KLD] output images, which is only defined behaviour inside a single object.
  constexpr int AT_KST = 32 * AT_KLD, AT_VST = 64 * AT_VLD;          // halfs per K / V^T stage
  __shared__ __attribute__((aligned(16))) _Float16 at_lds[2 * AT_KST + 2 * AT_VST];
  static_assert(4 * 32 * AT_KLD <= 2 * AT_KST + 2 * AT_VST, "the four per-wave output images must fit the K | V^T stages");
  _Float16 (*Ks)[AT_KST] = reinterpret_cast<_Float16 (*)[AT_KST]>(at_lds);
  _Float16 (*Vs)[AT_VST] = reinterpret_cast<_Float16 (*)[AT_VST]>(at_lds + 2 * AT_KST);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  // Workgroups go to the eight XCDs round-robin by linear id, and every XCD has its own L2: as a (query block, head, clip)
  // grid the twelve query blocks of one (clip, head) landed on eight different L2s and its K | V^T -- 384 KB that all
  // twelve read -- came from HBM up to eight times (rocprofv3 FETCH_SIZE, tiny, 64 clips: 1.28 GB per launch against
  // 221 MB of q | k | V^T).  So: linear id L -> XCD L % 8, and the query blocks of a group are consecutive slots of ONE
  // XCD: group = 8 (slot / nq) + xcd, query block = slot % nq (grid padded to whole groups of eight).
  const int nq = (T + 127) / 128;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = 8 * (slot / nq) + xcd;
  if (grp >= heads * n_clips) return;
  const int b = grp / heads, h = grp - b * heads;
  const int q0 = (slot % nq) * 128 + wave * 32;
  const long ld = 2L * D;
  const _Float16* Qp = qk + (long)b * T * ld + h * 64;
  const _Float16* Kp = Qp + D;
  const _Float16* Vp = vt + ((long)(b * heads + h) * 64) * ENC_TP;     // row d at Vp + d * ENC_TP
  const float sc = 0.125f * 1.4426950408889634f;    // 1 / sqrt(64) and log2(e): the exponentials are taken in base 2

  // staging roles: K tile = 32 rows x 8 chunks of 16 bytes, V^T tile = 64 rows x 4 chunks
  const int krow = tid >> 3, kch = tid & 7, vrow = tid >> 2, vch = tid & 3;
  const _Float16* kg = Kp + kch * 8;
  const _Float16* vg = Vp + (long)vrow * ENC_TP + vch * 8;
  auto kaddr = [&](int k0) { return kg + (long)min(k0 + krow, T - 1) * ld; };
  uint4 rk = *reinterpret_cast<const uint4*>(kaddr(0));
  uint4 rv = *reinterpret_cast<const uint4*>(vg);
  const int kso = krow * AT_KLD + kch * 8, vso = vrow * AT_VLD + vch * 8;
  *reinterpret_cast<uint4*>(&Ks[0][kso]) = rk;
  *reinterpret_cast<uint2*>(&Vs[0][vso]) = make_uint2(rv.x, rv.y);
  *reinterpret_cast<uint2*>(&Vs[0][vso + 4]) = make_uint2(rv.z, rv.w);

  // Q operand of k-step ks: Q[q = li][16 ks + 8 lh .. + 7]
  half8 qh[4];
  {
    const int q = min(q0 + li, T - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qh[ks] = *reinterpret_cast<const half8*>(Qp + (long)q * ld + 16 * ks + 8 * lh);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;      // m_run: integer, log2 units; l_run: this lane's keys only (halves joined at the end)
  __syncthreads();

  const int n_tiles = (T + 31) / 32;
  float m_fix = 0.f, inv_l = 0.f;         // NORM16: the row's maximum (log2 units) and 1 / sum over all keys
  if constexpr (NORM16) {
    float mr = -1e30f, lr = 0.f;
    for (int it = 0; it < n_tiles; ++it) {
      const int k0 = it * 32, buf = it & 1;
      rk = *reinterpret_cast<const uint4*>(kaddr(min(it + 1, n_tiles - 1) * 32));
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 kh = *reinterpret_cast<const half8*>(&Ks[buf][li * AT_KLD + 16 * ks + 8 * lh]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
      }
      if (k0 + 32 > T) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + acc_row_e(r, lane) >= T) s[r] = -3e38f;
      }
      float mloc = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mloc = fmaxf(fmaxf(mloc, s[r]), s[r + 1]);
      mloc = fmaxf(mloc, s[15]) * sc;
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float m_new = fmaxf(mr, mloc);
      lr *= __builtin_amdgcn_exp2f(mr - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_new));
      lr += psum;
      mr = m_new;
      __builtin_amdgcn_sched_barrier(0);
      *reinterpret_cast<uint4*>(&Ks[buf ^ 1][kso]) = rk;
      __syncthreads();
    }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lr), __float_as_uint(lr), false, false);
      lr = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    m_fix = mr;
    inv_l = 1.f / lr;
    // K tile 0 again (V^T tile 0 is still where the prologue put it); everyone is behind the loop's last barrier
    rk = *reinterpret_cast<const uint4*>(kaddr(0));
    *reinterpret_cast<uint4*>(&Ks[0][kso]) = rk;
    __syncthreads();
  }
  // ---- main pass.  Round 6 instruction diet (profiles/r06_attn_enc_pmc.json: the SIMD's vector issue port is 87 % busy, 107
  // vector instructions per tile and wave; every one that goes is ~0.8 % of the kernel), all of it bit-identical:
  //   * the loop is unrolled by the two staging buffers, so every LDS address is a register + an immediate (was: three
  //     v_add per tile and the scalar arithmetic that fed them);
  //   * the staging loads go through a wave-uniform base (scalar registers, stepped by scalar adds) + one 32-bit byte offset
  //     per thread, instead of a 64-bit pointer per thread stepped and clamped with vector instructions;
  //   * max(a, b) of two values the compiler cannot prove canonical (raw matrix-core results, the two halves of a lane swap)
  //     costs two canonicalising v_max x, x in front of the v_max; written as max(max(-3e38, a), b) it is ONE v_max3, which
  //     takes its operands as they are (twice per tile);
  //   * the tile's probability sum starts from its first pair instead of from 0.
  const char* Kbase = reinterpret_cast<const char*>(Kp);                // uniform in (clip, head)
  const char* Vbase = reinterpret_cast<const char*>(Vp);
  const unsigned kvo = (unsigned)((krow * (int)ld + kch * 8) * 2);      // this thread's piece of a K tile, from the tile's first row
  const int t_last = n_tiles - 1;
  const unsigned kvo_last = (unsigned)(((min(t_last * 32 + krow, T - 1) - t_last * 32) * (int)ld + kch * 8) * 2);   // rows past T - 1: the last row again
  const unsigned vvo = (unsigned)((vrow * ENC_TP + vch * 8) * 2);
  const long ktile_bytes = 32 * ld * 2;
  const float nlarge = -3.0e38f;          // below every score (masked keys are -3e38 too)
  auto tile = [&](auto bufc, int it) {
    constexpr int buf = decltype(bufc)::value;
    const int k0 = it * 32;
    {
      const int tn = min(it + 1, t_last);               // the last trip re-requests its own tile (never used)
      rk = *reinterpret_cast<const uint4*>(Kbase + (long)tn * ktile_bytes + (tn == t_last ? kvo_last : kvo));
      unsigned vo = vvo;
      asm volatile("" : "+v"(vo));                      // (or the loop-invariant Vbase + vvo is hoisted as a 64-bit vector pointer and stepped with vector adds)
      rv = *reinterpret_cast<const uint4*>(Vbase + (long)tn * 64 + vo);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T tile: rows = keys (li), columns = queries ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8 kh = *reinterpret_cast<const half8*>(&Ks[buf][li * AT_KLD + 16 * ks + 8 * lh]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
    }
    const bool tail = k0 + 32 > T;          // wave-uniform: the last tile has keys past the end
    if (tail) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + acc_row_e(r, lane) >= T) s[r] = -3e38f;
    }
    float mloc = nlarge;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mloc = fmaxf(fmaxf(mloc, s[r]), s[r + 1]);      // eight v_max3, the first against a constant
    mloc *= sc;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
      mloc = fmaxf(fmaxf(nlarge, __uint_as_float(sw[0])), __uint_as_float(sw[1]));
    }
    // Raise the reference exponent only when it is exceeded by more than 2^6 (p <= 128 fits f16 with room to spare),
    // and keep it an INTEGER: 2^(t - m) with integer m has the mantissa of 2^t, so the f16 rounding of a probability
    // does not depend on which tile last raised the reference -- the oracle rounds 2^(t - ceil(row max)) the same way.
    if (!NORM16 && __builtin_amdgcn_ballot_w64(mloc > m_run + 6.f) != 0ull) {
      const float m_new = ceilf(fmaxf(m_run, mloc));
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      m_run = m_new;
    }
    float psum = 0.f;
    half8 ph[2];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float p0, p1;
      if constexpr (NORM16) {
        p0 = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_fix)) * inv_l;
        p1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], sc, -m_fix)) * inv_l;
      } else {
        p0 = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_run));
        p1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], sc, -m_run));
      }
      psum = r == 0 ? p0 + p1 : psum + (p0 + p1);      // (0 + x == x exactly: the sum of round 5, one instruction shorter)
      const half2v pp = __builtin_convertvector(float2v{p0, p1}, half2v);
      ph[r >> 3][r & 7] = pp[0];
      ph[r >> 3][(r & 7) + 1] = pp[1];
    }
    l_run += psum;
    // ---- O^T += V^T . P^T: k-slot (lh, e) of step ks is the key of S^T register 8 ks + e, i.e. keys
    // k0 + 16 ks + 4 lh + {0..3} and + 8 + {0..3}: two 8-byte runs of row d (li, 32 + li) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const _Float16* r0 = &Vs[buf][li * AT_VLD + 16 * ks + 4 * lh];
      const _Float16* r1 = r0 + 32 * AT_VLD;
      half4 a0 = *reinterpret_cast<const half4*>(r0), a1 = *reinterpret_cast<const half4*>(r0 + 8);
      half4 b0 = *reinterpret_cast<const half4*>(r1), b1 = *reinterpret_cast<const half4*>(r1 + 8);
      if (tail) {   // V^T columns past the end hold whatever the buffer held: 0 x NaN must not reach the accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k0 + 16 * ks + 4 * lh + e >= T) { a0[e] = (_Float16)0.f; b0[e] = (_Float16)0.f; }
          if (k0 + 16 * ks + 4 * lh + 8 + e >= T) { a1[e] = (_Float16)0.f; b1[e] = (_Float16)0.f; }
        }
      }
      const half8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const half8 v1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, ph[ks], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, ph[ks], o1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<uint4*>(&Ks[buf ^ 1][kso]) = rk;
    *reinterpret_cast<uint2*>(&Vs[buf ^ 1][vso]) = make_uint2(rv.x, rv.y);
    *reinterpret_cast<uint2*>(&Vs[buf ^ 1][vso + 4]) = make_uint2(rv.z, rv.w);
    __syncthreads();
  };
  {
    int it = 0;
    for (; it + 1 < n_tiles; it += 2) {
      tile(std::integral_constant<int, 0>{}, it);
      tile(std::integral_constant<int, 1>{}, it + 1);
    }
    if (it < n_tiles) tile(std::integral_constant<int, 0>{}, it);
  }
  // ---- normalise; lane (li = query, lh) holds O[q][d] for d = acc_row(r) (+ 32 for o1): 8-byte stores ----
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = NORM16 ? 1.f : 1.f / l_run;       // NORM16: the probabilities were normalised before they were rounded
  // lane (li = query, lh) holds O[q][d] for d = acc_row(r) (+ 32 for o1).  Stored from here, every instruction would
  // touch 32 rows with 16 bytes each (16 instructions per wave, store-issue bound); through a [32][64 + 8] f16 image in
  // the K | V^T stages that are free now, a wave writes whole 128-byte rows, 8 rows per 16-byte-per-lane instruction.
  __syncthreads();                                   // every wave has read its last K / V^T tile
  _Float16* Timg = at_lds + wave * (32 * AT_KLD);     // 4 x 4608 bytes = the two K stages and the two V^T stages
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int d = 8 * g4 + 4 * lh;
    *reinterpret_cast<half4*>(Timg + li * AT_KLD + d) = to_half4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
    *reinterpret_cast<half4*>(Timg + li * AT_KLD + 32 + d) = to_half4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (lane >> 3) + 8 * p, c8 = (lane & 7) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(Timg + row * AT_KLD + c8);
    if (q0 + row < T) *reinterpret_cast<uint4*>(out + ((long)b * T + q0 + row) * D + h * 64 + c8) = v;
  }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (_Float16)src[i];
}

__global__ void f32_to_f16_rows_kernel(const float* __restrict__ src, long lds, _Float16* __restrict__ dst, long ldd,
                                       int cols, long rows) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  if (r < rows) dst[r * ldd + c] = (_Float16)src[r * lds + c];
}


// ---------------------------------------------------------------------------------------------
// The same loop over a 256 x 128 output tile: a wave owns 128 x 64 (acc[4][2]), so a trip around the k loop -- one
// vmcnt wait, one barrier, the LDS round trip of the first operands -- carries 16 MFMAs instead of 8, and the bytes
// requested per MFMA fall by a quarter (24 KB per 32-wide k-block for twice the outputs).  Ablations of the 128 x 128
// kernel on the 256-clip Whisper-base encoder had put the cost there: twice the MFMAs per trip +32 % of the GEMM time,
// every request served from L1 -13 %, LDS bank conflicts removed 0 % -- more than half of a trip is its fixed cost.
// 72 KB of LDS (three stages of 16 + 8 KB), 2 workgroups per CU.  Same operand layout, swizzle and epilogue (the
// wave's 128 rows go through the 32-row epilogue image as two 64-row halves).
// ---------------------------------------------------------------------------------------------
// MI = 32-row blocks per wave: 4 -> 256 x 128 tiles, 3 -> 192 x 128.  The shorter tile exists for the tail: 96 000 rows x
// 384 columns are 1 125 tiles of 256 rows on 512 workgroup slots = 2.2 rounds, i.e. three rounds with the last one a
// fifth full (out-projection, fc2, V^T); as 1 500 tiles of 192 rows they are 2.93 rounds of 3/4 the length (gemm_hh picks
// the height that minimises rounds x height).
constexpr int HD2_M = 256;      // the taller of the two
// One LDS-DMA request of 16 bytes per lane.  A plain (non-template) function on purpose: inside anything that depends on
// a template parameter -- a loop bound, a lambda of a kernel template -- the builtin fails to instantiate in hipcc's
// HOST pass, where it does not exist, and takes the enclosing kernel's stub with it without a diagnostic.
__device__ __forceinline__ void lds_dma16(const _Float16* src, unsigned char* dst) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  __builtin_amdgcn_global_load_lds(src, (lds_ptr)dst, 16, 0, 0);
}
template <int EM>      // EM = epilogue + 8 * MI (one parameter: hipcc's host pass leaves the stub of a two-parameter kernel template undefined)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_hd2_kernel(HGemmArgs g) {
  constexpr int EPI = EM & 7, MI = EM >> 3;
  constexpr int TM = 64 * MI, A_BYTES = TM * HH_K * 2, STAGE_BYTES = A_BYTES + HD_TILE_BYTES;     // 16 | 12 KB + 8 KB
  constexpr int NRD = 2 * MI + 4;                                                                   // operand reads per trip
  constexpr int RING_BYTES = HD_STAGES * STAGE_BYTES, IMG_BYTES = 4 * 2 * EP_IMAGE_BYTES;
  static_assert(MI == 3 || MI == 4, "192- or 256-row tiles");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING_BYTES > IMG_BYTES ? RING_BYTES : IMG_BYTES];   // [stage][A TM rows | W 128 rows]; then two epilogue images per wave
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_swizzle) {
    const int nt = gridDim.x;
    const int lin = blockIdx.y * nt + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    by = 8 * (slot / nt) + xcd;
    bx = slot % nt;
    if (by * TM >= g.M) return;
  }
  const int bz = blockIdx.z;
  const _Float16* __restrict__ A = g.A + (long)bz * g.strideA;
  const _Float16* __restrict__ W = g.W;
  const int m0 = by * TM, n0 = bx * HH_N;
  const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;

  // ---- requests: wave w fills rows [16 MI w, 16 MI (w + 1)) of A (MI 1 KB wave-requests) and [32 w, 32 w + 32) of W (two) ----
  const _Float16* ga[MI];
  const _Float16* gw[2];
#pragma unroll
  for (int u = 0; u < MI; ++u) {
    const int r = 16 * MI * wave + 16 * u + (lane >> 2);
    ga[u] = A + (long)min(m0 + r, g.M - 1) * g.lda + 8 * ((lane & 3) ^ ((r >> 2) & 3));   // clamped rows: results not stored
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = 32 * wave + 16 * u + (lane >> 2);
    gw[u] = W + (long)min(n0 + r, g.N - 1) * g.ldw + 8 * ((lane & 3) ^ ((r >> 2) & 3));
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto request = [&](int stage, int k0) {
    unsigned char* base = smem + stage * STAGE_BYTES;
    const long ka = a_seg_k(g, k0);
#pragma unroll
    for (int u = 0; u < MI; ++u) lds_dma16(ga[u] + ka, base + (wave * MI + u) * 1024);
#pragma unroll
    for (int u = 0; u < 2; ++u) lds_dma16(gw[u] + k0, base + A_BYTES + wave * 2048 + 1024 * u);
  };

  const int swz = (li >> 2) & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const unsigned a_ks0 = lds0 + (wm + li) * 64 + ((lh ^ swz) << 4);
  const unsigned a_ks1 = lds0 + (wm + li) * 64 + (((2 + lh) ^ swz) << 4);
  const unsigned w_ks0 = lds0 + A_BYTES + (wn + li) * 64 + ((lh ^ swz) << 4);
  const unsigned w_ks1 = lds0 + A_BYTES + (wn + li) * 64 + (((2 + lh) ^ swz) << 4);

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / HH_K;
  request(0, 0);
  if (nk > 1) request(1, HH_K);
  float bias[2];                             // epilogue operand, requested here (behind the first operand requests)
#pragma unroll
  for (int j = 0; j < 2; ++j) bias[j] = g.bias ? g.bias[min(n0 + wn + 32 * j + li, g.N - 1)] : 0.f;
  int stage = 0;
  for (int kb = 0; kb < nk; ++kb) {
    if (kb + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (MI + 2));      // vmcnt(MI + 2): the requests of stage kb + 1 may be pending
    else __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (kb + 2 < nk) {
      const int s2 = stage + 2 >= HD_STAGES ? stage + 2 - HD_STAGES : stage + 2;
      request(s2, (kb + 2) * HH_K);
    }
    const unsigned so = (unsigned)stage * STAGE_BYTES;
    half8 a0[4], a1[4], w0[2], w1[2];          // a0 / w0: k sub-step 0, a1 / w1: sub-step 1; index = 32-row block (a?[3]: MI = 4 only)
    asm volatile("ds_read_b128 %0, %1" : "=v"(a0[0]) : "v"(a_ks0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(w0[0]) : "v"(w_ks0 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w0[1]) : "v"(w_ks0 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a0[1]) : "v"(a_ks0 + so));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a0[2]) : "v"(a_ks0 + so));
    if constexpr (MI == 4) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a0[3]) : "v"(a_ks0 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(a1[0]) : "v"(a_ks1 + so));
    asm volatile("ds_read_b128 %0, %1" : "=v"(w1[0]) : "v"(w_ks1 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(w1[1]) : "v"(w_ks1 + so));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a1[1]) : "v"(a_ks1 + so));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a1[2]) : "v"(a_ks1 + so));
    if constexpr (MI == 4) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a1[3]) : "v"(a_ks1 + so));
#define HD2_MFMA(i_, j_, av, wv) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc[i_][j_], 0, 0, 0);
    // NRD reads in flight; the waits name how many may still be: after the first four, after the rest of sub-step 0, ...
    if constexpr (MI == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a0[0]), "+v"(w0[0]), "+v"(w0[1]), "+v"(a0[1]));
    else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[0]), "+v"(w0[0]), "+v"(w0[1]), "+v"(a0[1]));
    HD2_MFMA(0, 0, a0[0], w0[0]) HD2_MFMA(0, 1, a0[0], w0[1]) HD2_MFMA(1, 0, a0[1], w0[0]) HD2_MFMA(1, 1, a0[1], w0[1])
    if constexpr (MI == 4) {
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[2]), "+v"(a0[3]));
      HD2_MFMA(2, 0, a0[2], w0[0]) HD2_MFMA(2, 1, a0[2], w0[1]) HD2_MFMA(3, 0, a0[3], w0[0]) HD2_MFMA(3, 1, a0[3], w0[1])
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a1[0]), "+v"(w1[0]), "+v"(w1[1]), "+v"(a1[1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a0[2]));
      HD2_MFMA(2, 0, a0[2], w0[0]) HD2_MFMA(2, 1, a0[2], w0[1])
      asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a1[0]), "+v"(w1[0]), "+v"(w1[1]), "+v"(a1[1]));
    }
    HD2_MFMA(0, 0, a1[0], w1[0]) HD2_MFMA(0, 1, a1[0], w1[1]) HD2_MFMA(1, 0, a1[1], w1[0]) HD2_MFMA(1, 1, a1[1], w1[1])
    if constexpr (MI == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1[2]), "+v"(a1[3]));
      HD2_MFMA(2, 0, a1[2], w1[0]) HD2_MFMA(2, 1, a1[2], w1[1]) HD2_MFMA(3, 0, a1[3], w1[0]) HD2_MFMA(3, 1, a1[3], w1[1])
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1[2]));
      HD2_MFMA(2, 0, a1[2], w1[0]) HD2_MFMA(2, 1, a1[2], w1[1])
    }
#undef HD2_MFMA
    static_assert(NRD == 2 * MI + 4, "the lgkmcnt waits above count NRD reads");
    stage = stage + 1 == HD_STAGES ? 0 : stage + 1;
  }
  __builtin_amdgcn_s_barrier();          // every wave has read its last operands: the stages become epilogue images
  float* T0 = reinterpret_cast<float*>(smem + wave * (2 * EP_IMAGE_BYTES));
  hd2_epilogue<EPI, MI>(g, acc, T0, T0 + EP_IMAGE_BYTES / 4, m0, n0, wm, wn, lane, bz, bias);
}

}  // namespace

hipError_t layernorm_f16out(const float* x, const float* gamma, const float* beta, void* y, long rows, int D,
                            hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  _Float16* yh = reinterpret_cast<_Float16*>(y);
  switch (D) {
    case 384: hipLaunchKernelGGL(layernorm_h_kernel<6>, grid, block, 0, s, x, gamma, beta, yh, rows); break;
    case 512: hipLaunchKernelGGL(layernorm_h_kernel<8>, grid, block, 0, s, x, gamma, beta, yh, rows); break;
    case 768: hipLaunchKernelGGL(layernorm_h_kernel<12>, grid, block, 0, s, x, gamma, beta, yh, rows); break;
    case 1024: hipLaunchKernelGGL(layernorm_h_kernel<16>, grid, block, 0, s, x, gamma, beta, yh, rows); break;
    case 1280: hipLaunchKernelGGL(layernorm_h_kernel<20>, grid, block, 0, s, x, gamma, beta, yh, rows); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// K % 32 == 0, rows of A and W 16-byte aligned, N % 4 == 0 (the caller checks)
hipError_t gemm_hh(const HGemmArgs& g, int epi, int batch, hipStream_t s) {
  const int nt = (g.N + HH_N - 1) / HH_N, mt = (g.M + HH_M - 1) / HH_M;
  HGemmArgs a = g;
  dim3 grid(nt, mt, batch);
  if (a.xcd_swizzle) grid.y = (unsigned)(((mt + 7) / 8) * 8);      // whole groups of eight row tiles
  static const bool direct = [] { const char* e = dev_env("CRISPY_ASR_GEMM"); return !(e && e[0] == 'r'); }();   // "regs": the register-staged loop
  static const bool tall = [] { const char* e = dev_env("CRISPY_ASR_GEMM"); return !(e && e[0] == 's'); }();     // "square": 128 x 128 tiles only
  if ((direct || epi == EPI_KVH || epi == EPI_F32) && tall && g.M >= 4 * HD2_M) {      // 256 x 128 or 192 x 128 tiles
    // two workgroups per CU: rounds x tile height is what the launch costs; ties go to the taller tile
    // (per device: a process may hold handles on several devices -- ADVICE r4; both tile heights give the same bits,
    // tests/test_gpu_mode1.py::test_mode1_encoder_does_not_depend_on_the_gemm_tile_height)
    const int slots = [] {
      static int cu_of[64] = {};                 // 0 = not asked yet; races write the same value
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
      if (cu_of[dev] == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cu_of[dev] = n;
      }
      return 2 * cu_of[dev];
    }();
    static const int force_mi = [] { const char* e = test_env("CRISPY_ASR_TILE_ROWS"); return e ? std::atoi(e) / 64 : 0; }();   // 192 | 256
    auto cost = [&](int mi) { const long tiles = (long)((g.M + 64 * mi - 1) / (64 * mi)) * nt * batch; return ((tiles + slots - 1) / slots) * mi; };
    const int mi = (force_mi == 3 || force_mi == 4) ? force_mi : (cost(3) < cost(4) ? 3 : 4);
    const int mt2 = (g.M + 64 * mi - 1) / (64 * mi);
    dim3 grid2(nt, a.xcd_swizzle ? (unsigned)(((mt2 + 7) / 8) * 8) : (unsigned)mt2, batch);
#define HD2_LAUNCH(E) { if (mi == 3) hipLaunchKernelGGL(gemm_hd2_kernel<E + 8 * 3>, grid2, dim3(256), 0, s, a); else hipLaunchKernelGGL(gemm_hd2_kernel<E + 8 * 4>, grid2, dim3(256), 0, s, a); }
    switch (epi) {
      case EPI_F16: HD2_LAUNCH(EPI_F16) break;
      case EPI_RES: HD2_LAUNCH(EPI_RES) break;
      case EPI_VT: HD2_LAUNCH(EPI_VT) break;
      case EPI_TAB: HD2_LAUNCH(EPI_TAB) break;
      case EPI_KVH: HD2_LAUNCH(EPI_KVH) break;
      case EPI_F32: HD2_LAUNCH(EPI_F32) break;
      default: return hipErrorInvalidValue;
    }
#undef HD2_LAUNCH
    return hipGetLastError();
  }
  if (direct || epi == EPI_KVH || epi == EPI_F32) {
    switch (epi) {
      case EPI_F16: hipLaunchKernelGGL(gemm_hd_kernel<EPI_F16>, grid, dim3(256), 0, s, a); break;
      case EPI_RES: hipLaunchKernelGGL(gemm_hd_kernel<EPI_RES>, grid, dim3(256), 0, s, a); break;
      case EPI_VT: hipLaunchKernelGGL(gemm_hd_kernel<EPI_VT>, grid, dim3(256), 0, s, a); break;
      case EPI_TAB: hipLaunchKernelGGL(gemm_hd_kernel<EPI_TAB>, grid, dim3(256), 0, s, a); break;
      case EPI_KVH: hipLaunchKernelGGL(gemm_hd_kernel<EPI_KVH>, grid, dim3(256), 0, s, a); break;
      case EPI_F32: hipLaunchKernelGGL(gemm_hd_kernel<EPI_F32>, grid, dim3(256), 0, s, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (epi) {
    case EPI_F16: hipLaunchKernelGGL(gemm_hh_kernel<EPI_F16>, grid, dim3(256), 0, s, a); break;
    case EPI_RES: hipLaunchKernelGGL(gemm_hh_kernel<EPI_RES>, grid, dim3(256), 0, s, a); break;
    case EPI_VT: hipLaunchKernelGGL(gemm_hh_kernel<EPI_VT>, grid, dim3(256), 0, s, a); break;
    case EPI_TAB: hipLaunchKernelGGL(gemm_hh_kernel<EPI_TAB>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t attn_encoder_h(const void* qk, const void* vt, void* out, int B, int T, int D, int heads, hipStream_t s, int norm16) {
  const int nq = (T + 127) / 128, groups8 = (heads * B + 7) / 8;
  const dim3 grid((unsigned)(8 * groups8 * nq));
  if (norm16)
    hipLaunchKernelGGL(attn_enc_h_kernel<true>, grid, dim3(256), 0, s, reinterpret_cast<const _Float16*>(qk),
                       reinterpret_cast<const _Float16*>(vt), reinterpret_cast<_Float16*>(out), T, D, heads, B);
  else
    hipLaunchKernelGGL(attn_enc_h_kernel<false>, grid, dim3(256), 0, s, reinterpret_cast<const _Float16*>(qk),
                       reinterpret_cast<const _Float16*>(vt), reinterpret_cast<_Float16*>(out), T, D, heads, B);
  return hipGetLastError();
}

hipError_t convert_rows_f32_to_f16(const float* src, long lds, void* dst, long ldd, int cols, long rows, hipStream_t s) {
  const long n = rows * cols;
  hipLaunchKernelGGL(f32_to_f16_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, lds,
                     reinterpret_cast<_Float16*>(dst), ldd, cols, rows);
  return hipGetLastError();
}

hipError_t convert_f32_to_f16(const float* src, void* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                     reinterpret_cast<_Float16*>(dst), n);
  return hipGetLastError();
}

}  // namespace crispy
