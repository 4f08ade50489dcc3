// whisper_kernels.hip -- Whisper encoder/decoder building blocks for MI355X (gfx950).
//
// Replaces the whisper.cpp compute graph behind transcribe_rs::SpeechModel::transcribe
// (reference call sites src-tauri/src/managers/transcription.rs:183-185, 213-215);
// architecture: SURVEY.md Appendix B.2.
//
// Round-1 numerics: f32 end to end on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32
// products, f32 accumulation, 157 TFLOP/s peak) so that parity with the fp32 oracle is limited only by
// summation order.  bf16 operands are a later, separately validated step.
//
//   gemm_f32_nt_kernel     C[M,N] = A[M,K] . W[N,K]^T (+bias) (GELU) (+residual | +row-periodic table)
//                          128x128x16 tiles, 4 waves x (2x2) MFMA 32x32 tiles, LDS double buffer,
//                          A may be a strided *view* (lda < K): both Whisper convolutions run as GEMMs
//                          over overlapping rows of the frame-major activation without an im2col copy.
//   layernorm_kernel       one wave per row.
//   attn_enc_kernel        flash-style non-causal attention, one wave per 32 queries, S^T = K.Q^T so the
//                          softmax statistics are per lane and P^T feeds the P.V MFMAs from registers.
//   attn_dec_kernel        one wave per (clip, head): a single query against a KV cache / cross KV.
#include <algorithm>
#include <atomic>
#include "asr_common.h"
#include "asr_quant.h"

namespace crispy {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_M = 128, GB_N = 128, GB_K = 16, GB_LD = 20;  // LDS row stride 20 floats: conflict-free b128 reads

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// row index of accumulator register r for this lane (32x32 MFMA C/D layout)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_f32_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][GB_M * GB_LD];
  __shared__ __attribute__((aligned(16))) float Ws[2][GB_N * GB_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bz = blockIdx.z;
  const float* __restrict__ A = g.A + (long)bz * g.strideA;
  const float* __restrict__ W = g.W;
  float* __restrict__ C = g.C + (long)bz * g.strideC + (g.c_off_dev ? (long)(*g.c_off_dev) * g.c_off_scale : 0L);
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging: thread loads float4 #q of row (tid>>2) + 64*h, k-offset 4*(tid&3)
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  float4 ra[2], rw[2];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + lr + 64 * h, n = n0 + lr + 64 * h;
      ra[h] = (m < g.M) ? *reinterpret_cast<const float4*>(A + (long)m * g.lda + k0 + lk) : make_float4(0, 0, 0, 0);
      rw[h] = (n < g.N) ? *reinterpret_cast<const float4*>(W + (long)n * g.ldw + k0 + lk) : make_float4(0, 0, 0, 0);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(&As[buf][(lr + 64 * h) * GB_LD + lk]) = ra[h];
      *reinterpret_cast<float4*>(&Ws[buf][(lr + 64 * h) * GB_LD + lk]) = rw[h];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / GB_K;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kb = 0; kb < nk; ++kb) {
    const int buf = kb & 1;
    if (kb + 1 < nk) load_tiles((kb + 1) * GB_K);
    // operands: MFMA step s contracts the k-pair (s, s+8) of this 16-wide block; lane half lh owns k = 8*lh + s
    float a[2][8], w[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4* pa = reinterpret_cast<const float4*>(&As[buf][(wm + 32 * i + li) * GB_LD + 8 * lh]);
      const float4* pw = reinterpret_cast<const float4*>(&Ws[buf][(wn + 32 * i + li) * GB_LD + 8 * lh]);
      const float4 a0 = pa[0], a1 = pa[1], w0 = pw[0], w1 = pw[1];
      a[i][0] = a0.x; a[i][1] = a0.y; a[i][2] = a0.z; a[i][3] = a0.w;
      a[i][4] = a1.x; a[i][5] = a1.y; a[i][6] = a1.z; a[i][7] = a1.w;
      w[i][0] = w0.x; w[i][1] = w0.y; w[i][2] = w0.z; w[i][3] = w0.w;
      w[i][4] = w1.x; w[i][5] = w1.y; w[i][6] = w1.z; w[i][7] = w1.w;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], w[j][s], acc[i][j], 0, 0, 0);
    if (kb + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue.  The residual / row-table operands of a lane's 16 rows are requested together from clamped (always
  // valid) addresses before any of them is used: one exposed round trip per 32 x 32 block instead of one per element
  // (with a load, a wait and a store per row the short-K projections of the encoder spent their tail waiting on L2).
  const bool hm = g.hm_rows > 0;              // head-major store (cross K|V): see GemmArgs
  int hm_b0 = 0, hm_t0 = 0;
  if (hm) { hm_b0 = m0 / g.hm_rows; hm_t0 = m0 - hm_b0 * g.hm_rows; }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 32 * j + li;
      const int nc = min(n, g.N - 1);
      const float bias = g.bias ? g.bias[nc] : 0.f;
      long hm_col = 0;
      if (hm) {
        const int kv = nc >= g.hm_width ? 1 : 0, rem = nc - kv * g.hm_width;
        hm_col = ((long)kv * g.hm_width + (long)(rem >> 6) * 64) * g.hm_rows + (rem & 63);
      }
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {     // eight rows at a time: enough loads in flight, 3 waves per SIMD kept
        float extra[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) extra[r] = 0.f;
        if (g.residual) {
          const float* rp = g.residual + (long)bz * g.strideR + nc;
#pragma unroll
          for (int r = 0; r < 8; ++r)
            extra[r] = rp[(long)min(m0 + wm + 32 * i + acc_row(r0 + r, lane), g.M - 1) * g.ldr];
        }
        if (g.rowtab) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            extra[r] += g.rowtab[(long)(min(m0 + wm + 32 * i + acc_row(r0 + r, lane), g.M - 1) % g.rowtab_period) * g.N + nc];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int dm = wm + 32 * i + acc_row(r0 + r, lane);
          const int m = m0 + dm;
          float v = acc[i][j][r0 + r] + bias;
          if (g.gelu) v = g.gelu == 2 ? gelu_ggml(v) : gelu_erf(v);
          v += extra[r];
          if (m >= g.M || n >= g.N) continue;
          if (hm) {
            int t = hm_t0 + dm, b = hm_b0;      // a 128-row tile crosses at most one clip boundary (hm_rows >= 128)
            if (t >= g.hm_rows) { t -= g.hm_rows; ++b; }
            C[(long)b * g.N * g.hm_rows + (long)t * 64 + hm_col] = v;
          } else {
            C[(long)m * g.ldc + n] = v;
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Skinny GEMM for the decoder (M = clips of one decode step): latency-, not FLOP-bound.
// One workgroup per 32 clips x 32 output columns; the 4 waves split K four ways and feed the MFMA straight from
// global memory (both operands are K-contiguous rows, so a lane's operands are 16 contiguous floats per
// 32-wide K chunk: step s contracts k = s and k = s + 16); partial tiles are summed through LDS.
// No LDS staging, no barriers in the K loop.  The kernel is a chain of memory latencies, not of MFMAs (48 per wave
// at K = 384): three K chunks of operands are requested before the first MFMA, so Whisper-tiny's K = 384
// projections expose one round trip instead of three, and 32-row blocks put twice the workgroups on the chip
// (24 for N = 384).  W is re-read per row block from L2.
// ---------------------------------------------------------------------------------------------
constexpr int SK_PF = 3;                       // K chunks (32 wide) in flight per wave
typedef _Float16 sk_half8 __attribute__((ext_vector_type(8)));
constexpr int SK_WAVE_LDS = 2048;              // floats of LDS per wave (two 4 KB operand chunks; later the 32 x 33 partial tile)
__device__ __forceinline__ void sk_wave_sync() {     // LDS traffic of one wave: in order, so a fence for the compiler and a wait
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LN / GELU / RES are compile-time so that the epilogue operands (ln_s, ln_c or bias, residual) can be requested
// up front, next to the first K chunks, from clamped (always valid) addresses: with run-time flags every one of them
// sat behind its own branch in the epilogue and cost a serialized L2 round trip after the barrier.
// NW = waves of a workgroup = ways K is split.  4 for the K = d projections; 16 for the MLP's second GEMM (K = 4 d):
// with 4 waves that one ran four rounds of the memory latency on 24 workgroups (19.5 us at Whisper-tiny, three times
// the other projections), with 16 it runs one like the rest.
// WH: W is an f16 matrix (g.W reinterpreted, g.ldw in halves) and the products run on v_mfma_f32_32x32x16_f16 with
// the A operand rounded to f16 on the way in -- ggml's arithmetic for a plain mul_mat (precision mode 1: the attention
// output projections and the MLP's second GEMM, where no LayerNorm is folded in).  In f32 a 32 x 32 x 1536 tile is
// 768 MFMAs of 64 cycles on ONE CU: 5 us of matrix-pipe time for the K = 4 d projection at Whisper-tiny, whatever the
// number of clips; in f16 it is 96 MFMAs of 32 cycles.  Same loads for A, same epilogue; W arrives as two 16-byte
// pieces per lane and chunk (4 lanes per 64-byte row piece, slot = piece ^ ((row >> 2) & 3): see gemm_hd_kernel).
template <bool LN, bool GELU, bool RES, int NW, bool WH = false>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_f32_kernel(GemmArgs g) {
  static_assert(!(WH && LN), "the f16-weight form has no LayerNorm fold");
  extern __shared__ __attribute__((aligned(16))) float sk_smem[];
  // per wave 8 KB: the transposition area of its operand chunks (W, then A: 32 rows x 8 sixteen-byte pieces each); the
  // wave's partial tile replaces it after the K loop
  float (*red)[SK_WAVE_LDS] = reinterpret_cast<float (*)[SK_WAVE_LDS]>(sk_smem);
  float (*rstat)[32][2] = reinterpret_cast<float (*)[32][2]>(sk_smem + NW * SK_WAVE_LDS);
  constexpr int ET = NW >= 16 ? 1024 : NW >= 8 ? 512 : 256;    // threads that finish outputs (12 waves: the first 512)
  constexpr int RP = ET / 32, NQ = 32 / RP;    // epilogue: RP rows per pass, NQ passes
  constexpr int PF = NW > 12 ? 2 : SK_PF;            // 16 waves: 128 registers per lane, two chunks in flight (three spill 35 - 67 registers even with scalar bases)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int mb = blockIdx.y * 32;            // row block of 32 clips
  const float* __restrict__ A = g.A;
  const float* __restrict__ W = g.W;
  const bool second = g.C2 && n0 >= g.n_split;      // split output: a workgroup's 32 columns go to one destination
  const int kper = g.K / NW;                // K range of this wave (multiple of 32)
  const int kbeg = wave * kper;
  // Global loads are row-coalesced: 8 lanes take the 8 sixteen-byte pieces of one row's 32-wide K chunk (one 128-byte
  // line), an instruction covers 8 rows, four cover the chunk.  (Loading in MFMA operand order -- lane = row, 16
  // contiguous floats -- made every instruction touch 64 lines for 16 bytes each and the texture addresser, not the
  // memory latency, set the pace: 19.5 us for K = 1536 on 24 workgroups.)  The chunk goes through the wave's LDS area
  // to reach operand order; piece c of row r sits in slot c ^ (r & 7), so both the linear writes and the
  // row-per-lane 16-byte reads are conflict-free.  Operand registers, MFMA order and results are those of the
  // direct-load form.
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  // wave-uniform bases (scalar registers) + 32-bit byte offsets per lane: eight address registers instead of sixteen,
  // which is what lets the 16-wave form keep three chunks in flight inside its 128 registers
  const char* __restrict__ Wb = WH ? reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(W) + kbeg)
                                   : reinterpret_cast<const char*>(W + kbeg);
  const char* __restrict__ Ab = reinterpret_cast<const char*>(A + kbeg);
  unsigned wsrc[4], asrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (WH) {                                // j < 2: row 16 j + (lane >> 2), the 16-byte piece that belongs into slot lane & 3
      const int r = 16 * (j & 1) + (lane >> 2);
      wsrc[j] = (unsigned)(((long)min(n0 + r, g.N - 1) * g.ldw + 8 * ((lane & 3) ^ ((r >> 2) & 3))) * 2);
    } else {
      wsrc[j] = (unsigned)(((long)min(n0 + lr + 8 * j, g.N - 1) * g.ldw + 4 * lc) * 4);
    }
    asrc[j] = (unsigned)(((long)min(mb + lr + 8 * j, g.M - 1) * g.lda + 4 * lc) * 4);   // rows >= M: clamped row, never stored
  }
  f32x4* stw = reinterpret_cast<f32x4*>(sk_smem + wave * SK_WAVE_LDS);
  f32x4* sta = stw + 256;
  int rslot[4];                              // slots of this lane's operand pieces 4 lh + q of row li
#pragma unroll
  for (int q = 0; q < 4; ++q) rslot[q] = li * 8 + ((4 * lh + q) ^ (li & 7));
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float s0 = 0.f, q0 = 0.f;                 // row sum / sum of squares of this lane's share (LN folding)
  // the chunks in flight.  They are clang vectors, not HIP's float4: copies of the float4 struct between these
  // arrays and LDS kept all six arrays in scratch (400 bytes per lane), whatever the loop structure
  f32x4 w0[4], a0[4], w1[4], a1[4], w2[4], a2[4];
#define SK_REQUEST(WR, AR, KO)                                            \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                         \
    if (!WH || j < 2) WR[j] = *reinterpret_cast<const f32x4*>(Wb + (size_t)wsrc[j] + (WH ? 2 : 4) * (KO)); \
    AR[j] = *reinterpret_cast<const f32x4*>(Ab + (size_t)asrc[j] + 4 * (KO)); \
  }
  SK_REQUEST(w0, a0, 0)
  if (PF > 1 && 32 < kper) { SK_REQUEST(w1, a1, 32) }
  if (PF > 2 && 64 < kper) { SK_REQUEST(w2, a2, 64) }
  // epilogue operands of this thread's outputs: column ec of rows (tid >> 5) + 8 q
  const int ec = min(tid & 31, g.N - 1 - n0), enn = n0 + ec;
  float e_s = 0.f, e_c = 0.f, e_res[NQ];
  // The LayerNorm-fold forms with 12 / 16 waves (128 / 168 registers per lane) have no room to keep these three to six
  // values across the K loop: the compiler parked them in scratch (a store behind a vmcnt wait in the prologue, a
  // reload -- one more memory round trip -- after the loop).  Those forms request them AFTER the loop instead, under
  // the LDS hand-off and the barrier: the same round trip, no scratch (tests/test_build_resources.py).
  constexpr bool LATE_EPI = LN && (NW >= 16 || (NW >= 12 && RES));
  auto request_epilogue = [&]() {
    if (LN) { e_s = g.ln_s[enn]; e_c = g.ln_c[enn]; }
    else if (g.bias) e_c = g.bias[enn];
    if (RES) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) e_res[q] = g.residual[(long)min(mb + min((tid >> 5) + RP * q, 31), g.M - 1) * g.ldr + enn];
    }
  };
  if (!LATE_EPI) request_epilogue();
  const long coff = g.c_off_dev ? (long)(*g.c_off_dev) * g.c_off_scale : 0L;
  float* __restrict__ C = second ? g.C2 + coff - g.n_split : g.C + (g.C2 ? 0L : coff);
  const long ldc = second ? g.ldc2 : g.ldc;
  // one chunk: registers -> LDS, request the chunk PF ahead into the same registers, LDS -> operand order, 16 MFMAs
#define SK_CHUNK(WR, AR, KC)                                                                        \
  {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { if (!WH || j < 2) stw[64 * j + lane] = WR[j]; sta[64 * j + lane] = AR[j]; } \
    if ((KC) + 32 * PF < kper) { SK_REQUEST(WR, AR, (KC) + 32 * PF) }                               \
    sk_wave_sync();                                                                                 \
    f32x4 cw[4], ca[4];                                                                            \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) { if (!WH) cw[q] = stw[rslot[q]]; ca[q] = sta[rslot[q]]; } \
    if (WH) { cw[0] = stw[li * 4 + ((2 * lh) ^ ((li >> 2) & 3))]; cw[1] = stw[li * 4 + ((2 * lh + 1) ^ ((li >> 2) & 3))]; } \
    sk_wave_sync();                                                                                 \
    if (WH) {                                                                                       \
      _Pragma("unroll") for (int st = 0; st < 2; ++st) {        /* k = 16 lh + 8 st + e for both operands */ \
        const sk_half8 av = {(_Float16)ca[2 * st].x, (_Float16)ca[2 * st].y, (_Float16)ca[2 * st].z, (_Float16)ca[2 * st].w, \
                             (_Float16)ca[2 * st + 1].x, (_Float16)ca[2 * st + 1].y, (_Float16)ca[2 * st + 1].z, (_Float16)ca[2 * st + 1].w}; \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(sk_half8, cw[st]), acc, 0, 0, 0); \
      }                                                                                             \
    } else {                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
      const float wv[4] = {cw[q].x, cw[q].y, cw[q].z, cw[q].w};                                     \
      const float xv[4] = {ca[q].x, ca[q].y, ca[q].z, ca[q].w};                                     \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[e], wv[e], acc, 0, 0, 0);                     \
        if (LN) { s0 += xv[e]; q0 = fmaf(xv[e], xv[e], q0); }                                       \
      }                                                                                             \
    }                                                                                               \
    }                                                                                               \
  }
  for (int kc0 = 0; kc0 < kper; kc0 += 32 * PF) {
    SK_CHUNK(w0, a0, kc0)
    if (PF > 1 && kc0 + 32 < kper) SK_CHUNK(w1, a1, kc0 + 32)
    if (PF > 2 && kc0 + 64 < kper) SK_CHUNK(w2, a2, kc0 + 64)
  }
#undef SK_CHUNK
#undef SK_REQUEST
  if (LATE_EPI) request_epilogue();
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][acc_row(r, lane) * 33 + li] = acc[r];     // own area: after the wave's last reads
  if (LN) {
    // the two half-waves hold the two 16-wide halves of every 32-wide K chunk of row li
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    if (lh == 0) { rstat[wave][li][0] = s0; rstat[wave][li][1] = q0; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ml = min((tid >> 5) + RP * q, 31);   // row inside this 32-row block (threads >= ET: clamped, not stored)
    const int m = mb + ml;
    float v = red[0][ml * 33 + ec];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) v += red[w2][ml * 33 + ec];      // fixed order: wave 0, 1, ...
    if (LN) {
      float sum = rstat[0][ml][0], sq = rstat[0][ml][1];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) { sum += rstat[w2][ml][0]; sq += rstat[w2][ml][1]; }
      const float mean = sum / (float)g.K;
      const float var = fmaxf(sq / (float)g.K - mean * mean, 0.f);
      const float rstd = 1.f / sqrtf(var + 1e-5f);
      v = rstd * (v - mean * e_s) + e_c;
    } else {
      v += e_c;
    }
    if (GELU) v = g.gelu == 2 ? gelu_ggml(v) : gelu_erf(v);
    if (RES) v += e_res[q];
    if (tid < ET && m < g.M && n0 + (tid & 31) < g.N) {                             // only the stores are predicated
      if (second && g.c2_half) (reinterpret_cast<_Float16*>(g.C2) + coff - g.n_split)[(long)m * ldc + enn] = (_Float16)v;
      else C[(long)m * ldc + enn] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The skinny kernel over RESIDENT QUANTISED weights (crispy_asr_load_resident; asr_quant.h).  Same decomposition, same
// A path (row-coalesced loads, transposed through the wave's LDS area), same MFMA order, same epilogue -- the W operand
// is different: a 32-wide K chunk of one output row is exactly ONE ggml block, and the 16 k values lane (row li, half
// lh) contracts are the block's low (lh = 0) or high (lh = 1) nibbles, so every lane fetches its row's block (5 - 6
// dwords, 18 - 24 bytes; two lanes share a block) straight into registers and de-quantises its 16 values there:
// no LDS round trip for W, 0.56 - 1.06 bytes per weight from HBM instead of 2 or 4, and no separate de-quantisation
// launch.  Values are computed with the loader's operations in the loader's order (int -> float, one multiply, one add,
// times gamma[k] for the LayerNorm-folded form -- fold_ln's W' = W . diag(gamma) -- each rounded on its own; then
// rounded to f16 for the WH form), so results equal those of the dense kernels on the de-quantised weights bit for bit.
// ---------------------------------------------------------------------------------------------
template <int TT> struct QFetch {
  static constexpr int NWD = TT == QT_Q8_0 ? 5 : (quant_block_bytes(TT) + 3) / 4;
  unsigned w[NWD];
};
template <int TT>
__device__ __forceinline__ void q_fetch(QFetch<TT>& f, const unsigned char* b, int lh) {
  if (TT == QT_Q8_0) {
    unsigned short dv;
    __builtin_memcpy(&dv, b, 2);
    f.w[0] = dv;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.w[1 + i] = q_u32(b + 2 + 16 * lh + 4 * i);
  } else {
#pragma unroll
    for (int i = 0; i < QFetch<TT>::NWD; ++i) f.w[i] = q_u32(b + 4 * i);      // (the last dword may reach 2 bytes past the block)
  }
}
__device__ __forceinline__ float q_half_bits(unsigned v) {
  const unsigned short u = (unsigned short)(v & 0xffffu);
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}
// y[s] = weight 16 lh + s of the fetched block, s < 16
template <int TT>
__device__ __forceinline__ void q_half_block(const QFetch<TT>& f, int lh, float (&y)[16]) {
#pragma clang fp contract(off)
  const float d = q_half_bits(f.w[0]);
  if (TT == QT_Q8_0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) y[4 * i + e] = (float)(int)(signed char)((f.w[1 + i] >> (8 * e)) & 0xff) * d;
    return;
  }
  constexpr bool has_m = TT == QT_Q4_1 || TT == QT_Q5_1, has_h = TT == QT_Q5_0 || TT == QT_Q5_1;
  const float m = has_m ? q_half_bits(f.w[0] >> 16) : 0.f;
  unsigned qh = 0u, qs[4];
  if (TT == QT_Q4_0) {          // {d}{qs 16} from byte 2
#pragma unroll
    for (int i = 0; i < 4; ++i) qs[i] = (f.w[i] >> 16) | (f.w[i + 1] << 16);
  } else if (TT == QT_Q4_1) {   // {d, m}{qs} from byte 4
#pragma unroll
    for (int i = 0; i < 4; ++i) qs[i] = f.w[1 + i];
  } else if (TT == QT_Q5_0) {   // {d}{qh 4}{qs} : qh at byte 2, qs at byte 6
    qh = (f.w[0] >> 16) | (f.w[1] << 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) qs[i] = (f.w[1 + i] >> 16) | (f.w[2 + i] << 16);
  } else {                      // q5_1 {d, m}{qh}{qs}
    qh = f.w[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) qs[i] = f.w[2 + i];
  }
  const int sh = 4 * lh;
  const unsigned hb = has_h ? (qh >> (16 * lh)) : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * i + e;
      int x = (int)((qs[i] >> (8 * e + sh)) & 0x0fu);
      if (has_h) x |= (int)((hb >> j) & 1u) << 4;
      if (TT == QT_Q4_0) y[j] = (float)(x - 8) * d;
      else if (TT == QT_Q5_0) y[j] = (float)(x - 16) * d;
      else y[j] = (float)x * d + m;
    }
}

template <int TT, bool LN, bool GELU, bool RES, int NW, bool WH>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_q_kernel(GemmArgs g) {
  static_assert(!(WH && LN), "the f16 form has no LayerNorm fold");
  extern __shared__ __attribute__((aligned(16))) float sk_smem[];
  float (*red)[SK_WAVE_LDS] = reinterpret_cast<float (*)[SK_WAVE_LDS]>(sk_smem);
  float (*rstat)[32][2] = reinterpret_cast<float (*)[32][2]>(sk_smem + NW * SK_WAVE_LDS);
  constexpr int ET = NW >= 16 ? 1024 : NW >= 8 ? 512 : 256;
  constexpr int RP = ET / 32, NQ = 32 / RP;
  // chunks in flight: A 16 + block 6 (+ gamma 16) registers each; the LayerNorm-folded form with 16 waves has 128 registers
  // per lane and keeps one (its K is 512 or 1024: one or two chunks per wave) -- two spilled 72 - 104 bytes
  constexpr int PF = (LN && NW >= 16) ? 1 : 2;
  constexpr int BB = quant_block_bytes(TT);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int mb = blockIdx.y * 32;
  const float* __restrict__ A = g.A;
  const bool second = g.C2 && n0 >= g.n_split;
  const int kper = g.K / NW;
  const int kbeg = wave * kper;
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  const char* __restrict__ Ab = reinterpret_cast<const char*>(A + kbeg);
  unsigned asrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) asrc[j] = (unsigned)(((long)min(mb + lr + 8 * j, g.M - 1) * g.lda + 4 * lc) * 4);
  // this lane's weight row: part p of the row-fused matrix, local row r -> its blocks are contiguous
  const unsigned char* __restrict__ wrow;
  {
    const int row = min(n0 + li, g.N - 1);
    const int p = row / g.wq_rows;
    const unsigned char* base = p == 0 ? g.wq[0] : (p == 1 ? g.wq[1] : g.wq[2]);
    wrow = base + ((long)(row - p * g.wq_rows) * (g.K / 32) + kbeg / 32) * BB;
  }
  const float* __restrict__ gam = LN ? g.wq_gamma + kbeg + 16 * lh : nullptr;
  f32x4* sta = reinterpret_cast<f32x4*>(sk_smem + wave * SK_WAVE_LDS) + 256;
  int rslot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rslot[q] = li * 8 + ((4 * lh + q) ^ (li & 7));
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float s0 = 0.f, q0 = 0.f;
  f32x4 a0[4], a1[4], g0[4], g1[4];
  QFetch<TT> b0, b1;
#define SKQ_REQUEST(BR, AR, GR, KO)                                                     \
  {                                                                                     \
    q_fetch<TT>(BR, wrow + (long)((KO) / 32) * BB, lh);                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
      AR[j] = *reinterpret_cast<const f32x4*>(Ab + (size_t)asrc[j] + 4 * (KO));         \
      if (LN) GR[j] = *reinterpret_cast<const f32x4*>(gam + (KO) + 4 * j);              \
    }                                                                                   \
  }
  SKQ_REQUEST(b0, a0, g0, 0)
  if (PF > 1 && 32 < kper) SKQ_REQUEST(b1, a1, g1, 32)
  const int ec = min(tid & 31, g.N - 1 - n0), enn = n0 + ec;
  float e_s = 0.f, e_c = 0.f, e_res[NQ];
  auto request_epilogue = [&]() {
    if (LN) { e_s = g.ln_s[enn]; e_c = g.ln_c[enn]; }
    else if (g.bias) e_c = g.bias[enn];
    if (RES) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) e_res[q] = g.residual[(long)min(mb + min((tid >> 5) + RP * q, 31), g.M - 1) * g.ldr + enn];
    }
  };
  const long coff = g.c_off_dev ? (long)(*g.c_off_dev) * g.c_off_scale : 0L;
  float* __restrict__ C = second ? g.C2 + coff - g.n_split : g.C + (g.C2 ? 0L : coff);
  const long ldc = second ? g.ldc2 : g.ldc;
#define SKQ_CHUNK(BR, AR, GR, KC)                                                                   \
  {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) sta[64 * j + lane] = AR[j];                       \
    float wy[16];                                                                                   \
    q_half_block<TT>(BR, lh, wy);                                                                   \
    if (LN) {                                                                                       \
      _Pragma("clang fp contract(off)")                                                             \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        wy[4 * j] = wy[4 * j] * GR[j].x; wy[4 * j + 1] = wy[4 * j + 1] * GR[j].y;                   \
        wy[4 * j + 2] = wy[4 * j + 2] * GR[j].z; wy[4 * j + 3] = wy[4 * j + 3] * GR[j].w;           \
      }                                                                                             \
    }                                                                                               \
    if ((KC) + 32 * PF < kper) SKQ_REQUEST(BR, AR, GR, (KC) + 32 * PF)                              \
    sk_wave_sync();                                                                                 \
    f32x4 ca[4];                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) ca[q] = sta[rslot[q]];                            \
    sk_wave_sync();                                                                                 \
    if (WH) {                                                                                       \
      _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                            \
        const sk_half8 av = {(_Float16)ca[2 * st].x, (_Float16)ca[2 * st].y, (_Float16)ca[2 * st].z, (_Float16)ca[2 * st].w, \
                             (_Float16)ca[2 * st + 1].x, (_Float16)ca[2 * st + 1].y, (_Float16)ca[2 * st + 1].z, (_Float16)ca[2 * st + 1].w}; \
        const sk_half8 wv = {(_Float16)wy[8 * st], (_Float16)wy[8 * st + 1], (_Float16)wy[8 * st + 2], (_Float16)wy[8 * st + 3], \
                             (_Float16)wy[8 * st + 4], (_Float16)wy[8 * st + 5], (_Float16)wy[8 * st + 6], (_Float16)wy[8 * st + 7]}; \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc, 0, 0, 0);                         \
      }                                                                                             \
    } else {                                                                                        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
        const float xv[4] = {ca[q].x, ca[q].y, ca[q].z, ca[q].w};                                   \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[e], wy[4 * q + e], acc, 0, 0, 0);           \
          if (LN) { s0 += xv[e]; q0 = fmaf(xv[e], xv[e], q0); }                                     \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
  }
  for (int kc0 = 0; kc0 < kper; kc0 += 32 * PF) {
    SKQ_CHUNK(b0, a0, g0, kc0)
    if (PF > 1 && kc0 + 32 < kper) SKQ_CHUNK(b1, a1, g1, kc0 + 32)
  }
#undef SKQ_CHUNK
#undef SKQ_REQUEST
  request_epilogue();
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][acc_row(r, lane) * 33 + li] = acc[r];
  if (LN) {
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    if (lh == 0) { rstat[wave][li][0] = s0; rstat[wave][li][1] = q0; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ml = min((tid >> 5) + RP * q, 31);
    const int m = mb + ml;
    float v = red[0][ml * 33 + ec];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) v += red[w2][ml * 33 + ec];
    if (LN) {
      float sum = rstat[0][ml][0], sq = rstat[0][ml][1];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) { sum += rstat[w2][ml][0]; sq += rstat[w2][ml][1]; }
      const float mean = sum / (float)g.K;
      const float var = fmaxf(sq / (float)g.K - mean * mean, 0.f);
      const float rstd = 1.f / sqrtf(var + 1e-5f);
      v = rstd * (v - mean * e_s) + e_c;
    } else {
      v += e_c;
    }
    if (GELU) v = g.gelu == 2 ? gelu_ggml(v) : gelu_erf(v);
    if (RES) v += e_res[q];
    if (tid < ET && m < g.M && n0 + (tid & 31) < g.N) {
      if (second && g.c2_half) (reinterpret_cast<_Float16*>(g.C2) + coff - g.n_split)[(long)m * ldc + enn] = (_Float16)v;
      else C[(long)m * ldc + enn] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (eps 1e-5), one wave per row; D <= 1280, multiple of 64
// ---------------------------------------------------------------------------------------------
template <int PER>     // PER = D / 64 columns per lane: compile-time, so the row is requested in one go (no branch per load)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
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
  float* yr = y + row * D;
#pragma unroll
  for (int q = 0; q < PER; ++q) yr[lane + 64 * q] = (v[q] - mean) * rstd * gm[q] + bt[q];
}

// ---------------------------------------------------------------------------------------------
// Encoder self-attention (non-causal), head dim 64.  qkv: [B][T][3*D] with q | k | v column blocks.
// grid (ceil(T/128), heads, B), 4 waves, wave = 32 queries.
//   S^T[key][q] = sum_d K[key][d] Q[q][d]    (A = K rows, B = Q^T; MFMA step s contracts d = s and s+32)
//   O^T[d][q]  += sum_key V[key][d] P^T[key][q]   (A = V^T, B = P^T straight from the S^T accumulator:
//                register r of lane half h is key (r&3) + 8(r>>2) + 4h, exactly the k-pair of step r)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_enc_kernel(const float* __restrict__ qkv, float* __restrict__ out,
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

  // Q operand: lane (q = li, half lh) holds Q[q][32*lh + s], s = 0..31, pre-scaled by 1/8 (exact)
  float qreg[32];
  {
    const int q = min(q0 + li, T - 1);
    const float4* p = reinterpret_cast<const float4*>(Qp + (long)q * ld + 32 * lh);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 t = p[c];
      qreg[4 * c] = t.x * 0.125f; qreg[4 * c + 1] = t.y * 0.125f;
      qreg[4 * c + 2] = t.z * 0.125f; qreg[4 * c + 3] = t.w * 0.125f;
    }
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  // K rows of the next tile are loaded while the current tile's P.V MFMAs run, V rows before the softmax arithmetic:
  // with two waves per SIMD an exposed HBM/L2 round trip per product left the matrix pipe 40 % idle
  float kreg[32];
  auto load_k = [&](int k0) {
    const int key = min(k0 + li, T - 1);
    const float4* p = reinterpret_cast<const float4*>(Kp + (long)key * ld + 32 * lh);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 t = p[c];
      kreg[4 * c] = t.x; kreg[4 * c + 1] = t.y; kreg[4 * c + 2] = t.z; kreg[4 * c + 3] = t.w;
    }
  };
  load_k(0);
  for (int k0 = 0; k0 < T; k0 += 32) {
    // ---- S^T tile ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[st], qreg[st], s, 0, 0, 0);
    // V rows of this tile and K rows of the next one go out now
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(k0 + acc_row(r, lane), T - 1);
      const float* vp = Vp + (long)key * ld;
      v0[r] = vp[li];
      v1[r] = vp[32 + li];
    }
    if (k0 + 32 < T) load_k(k0 + 32);
    // ---- online softmax over this lane's 16 keys + the partner half's 16 ----
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, lane);
      if (key >= T) s[r] = -1e30f;
      mloc = fmaxf(mloc, s[r]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[r], s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[r], s[r], o1, 0, 0, 0);
    }
  }
  // ---- normalise, transpose through LDS, store rows of 64 floats ----
  const float inv = 1.f / l_run;
  float* st = stage[wave];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = acc_row(r, lane);
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

// ---------------------------------------------------------------------------------------------
// Decoder attention for ONE query per (clip, head): scores against n_keys cached keys, softmax, P.V.
// q: [B][D] (row stride ldq), K/V: [B][n_ctx][ldkv] with this head's 64 columns at koff/voff + h*64.
// grid (heads, B), 4 waves: the keys are split four ways (flash-decoding style) and the partial
// (max, sum, P.V) triples are merged through LDS.  n_keys = n_keys_base + *pos_dev (graph replay keeps
// the launch arguments fixed while the position advances on the device).  Memory-bound on K/V.
// ---------------------------------------------------------------------------------------------
#ifndef AD_WAVES_N
#define AD_WAVES_N 16
#endif
constexpr int AD_WAVES = AD_WAVES_N;   // key partitions per (clip, head): 1.2 GB of cross K|V per step want many loads in flight
// NT: non-temporal request (the K|V stream of a decode step over many clips, see attn_dec_x16_kernel)
template <class KV, bool NT> __device__ __forceinline__ float4 ld4(const KV* p);
template <> __device__ __forceinline__ float4 ld4<float, false>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<float, true>(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ float4 ld4<_Float16, false>(const _Float16* p) {
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  const half4 h = *reinterpret_cast<const half4*>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <> __device__ __forceinline__ float4 ld4<_Float16, true>(const _Float16* p) {
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  const half4 h = __builtin_nontemporal_load(reinterpret_cast<const half4*>(p));
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
// KV = float (self-attention cache, default cross K|V) or _Float16 (cross K|V in precision mode 1: half the bytes of
// the stream that dominates a decode step; whisper.cpp keeps its KV caches in f16 as well)
template <class KV, bool STREAM_KV = false>
__global__ __launch_bounds__(64 * AD_WAVES) void attn_dec_kernel(const float* __restrict__ q, long ldq,
                                                       const KV* __restrict__ kv, long kv_batch_stride,
                                                       long ldkv, long head_stride, long koff, long voff, int n_keys_base,
                                                       const int* __restrict__ pos_dev, float* __restrict__ out, long ldo,
                                                       AttnRows rows) {
  __shared__ float p_s[1536];
  __shared__ __attribute__((aligned(16))) float q_s[64];
  __shared__ __attribute__((aligned(16))) float part_o[AD_WAVES][64];
  __shared__ float part_m[AD_WAVES], part_l[AD_WAVES];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, b = blockIdx.y;
  // rows.group query rows per clip (asr_common.h: AttnRows): row b belongs to clip b / group and reads that clip's K|V
  const int clip = b / rows.group;
  // rows.key_off: the clip's keys start at cache row key_off[clip] (left-padded prompts); the partition below then runs
  // over the same key COUNT from a shifted base -- the arithmetic of the un-padded clip, bit for bit
  const int k_off = rows.key_off ? rows.key_off[clip] : 0;
  const int n_keys = n_keys_base + (pos_dev ? *pos_dev : 0) + rows.key_step * (b % rows.group) - k_off;
  if (n_keys <= 0) {                    // a padding row in front of the clip's prompt: nothing to attend to, never read
    if (tid < 64) out[(long)b * ldo + h * 64 + tid] = 0.f;
    return;
  }
  if (tid < 64) q_s[tid] = q[(long)b * ldq + h * 64 + tid] * 0.125f;
  __syncthreads();
  const KV* Kb = kv + (long)clip * kv_batch_stride + koff + h * head_stride + (long)k_off * ldkv;
  const KV* Vb = kv + (long)clip * kv_batch_stride + voff + h * head_stride + (long)k_off * ldkv;
  const int per = (n_keys + AD_WAVES - 1) / AD_WAVES;
  const int k_lo = wave * per, k_hi = min(n_keys, k_lo + per);
  // scores: 16 lanes share one key row (coalesced 256-byte reads, 4 keys per wave instruction), the
  // 16 partial dot products are summed inside the DPP row
  float mloc = -1e30f;
  {
    const int c = lane & 15, sub = lane >> 4;
    const float4 qv = *reinterpret_cast<const float4*>(&q_s[4 * c]);
    constexpr int SU = 8;                                  // independent 1-KB loads in flight per wave
    for (int kb = k_lo; kb < k_hi; kb += 4 * SU) {
      float sacc[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int k = kb + 4 * u + sub;
        sacc[u] = 0.f;
        if (k < k_hi) {
          const float4 t = ld4<KV, STREAM_KV>(Kb + (long)k * ldkv + 4 * c);
          sacc[u] = t.x * qv.x;
          sacc[u] = fmaf(t.y, qv.y, sacc[u]);
          sacc[u] = fmaf(t.z, qv.z, sacc[u]);
          sacc[u] = fmaf(t.w, qv.w, sacc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        float v = sacc[u];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));
        const int k = kb + 4 * u + sub;
        if (k < k_hi) {
          if (c == 0) p_s[k] = v;
          mloc = fmaxf(mloc, v);
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
  float lsum = 0.f;
  for (int k = k_lo + lane; k < k_hi; k += 64) {
    const float p = __expf(p_s[k] - mloc);
    p_s[k] = p;
    lsum += p;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // P.V with the same access shape as the scores: 16 lanes share one 256-byte V row (4 dims each), 4 rows per wave
  // instruction, SU instructions (8 KB) in flight per wave -- the one-float-per-lane version kept 2 KB in flight and
  // spent most of the kernel waiting on it.  The four key residues are summed across the DPP rows at the end.
  {
    const int c = lane & 15, sub = lane >> 4;
    constexpr int SU = 8;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = k_lo; kb < k_hi; kb += 4 * SU) {
      float4 t[SU];
      float p[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int k = kb + 4 * u + sub;
        t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        p[u] = 0.f;
        if (k < k_hi) {
          t[u] = ld4<KV, STREAM_KV>(Vb + (long)k * ldkv + 4 * c);
          p[u] = p_s[k];
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        acc.x = fmaf(p[u], t[u].x, acc.x);
        acc.y = fmaf(p[u], t[u].y, acc.y);
        acc.z = fmaf(p[u], t[u].z, acc.z);
        acc.w = fmaf(p[u], t[u].w, acc.w);
      }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off, 64);
      acc.y += __shfl_xor(acc.y, off, 64);
      acc.z += __shfl_xor(acc.z, off, 64);
      acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (sub == 0) *reinterpret_cast<float4*>(&part_o[wave][4 * c]) = acc;
  }
  if (lane == 0) { part_m[wave] = mloc; part_l[wave] = lsum; }
  __syncthreads();
  if (wave == 0) {
    float m = part_m[0];
#pragma unroll
    for (int w = 1; w < AD_WAVES; ++w) m = fmaxf(m, part_m[w]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < AD_WAVES; ++w) {
      const float sc = __expf(part_m[w] - m);   // empty partitions have m = -1e30 -> scale 0
      o = fmaf(part_o[w][lane], sc, o);
      l = fmaf(part_l[w], sc, l);
    }
    out[(long)b * ldo + h * 64 + lane] = o / l;
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-attention of a decode step over an f16 K|V (precision mode 1), <= 1536 keys: every byte is requested before
// anything is computed.  A (clip, head) has 1500 keys x 128 bytes of K and as much of V; its 16 waves take 94 keys
// each -- 12 KB of K and 12 KB of V, which is 48 + 48 registers per lane.  So a wave issues its 24 sixteen-byte loads
// per lane back to back (8 lanes per key row, 8 keys per instruction: 1 KB per instruction), and the kernel is one
// round trip to HBM with the whole K|V of the layer in flight, instead of the generic kernel's K pass, softmax, V pass
// with 4 KB per wave in flight.  V does not wait for the scores; nothing goes through LDS but the 16 partial results.
// Same partitioning and merge as attn_dec_kernel (per-wave max / sum / P.V merged by wave 0).
// ---------------------------------------------------------------------------------------------
// ADX_SLOTS = key slots of 8 keys per wave: 12 for the cross-attention (16 waves x 96 keys >= 1536); 1 / 2 / 4 for the
// self-attention over the f16 K|V cache of mode 1 (<= 128 / 256 / 512 positions; the bound is known when the step is
// captured, the key count itself comes from the device counter).
template <int ADX_SLOTS, bool STREAM_KV>
__global__ __launch_bounds__(64 * AD_WAVES) void attn_dec_x16_kernel(const float* __restrict__ q, long ldq,
                                                       const _Float16* __restrict__ kv, long kv_batch_stride,
                                                       long ldkv, long head_stride, long koff, long voff, int n_keys_base,
                                                       const int* __restrict__ pos_dev, float* __restrict__ out, long ldo,
                                                       AttnRows rows) {
  // rows.group query rows per clip (asr_common.h: AttnRows): row b belongs to clip b / group and reads that clip's K|V
  const int b = blockIdx.y;
  const int clip = b / rows.group;
  const int k_off = rows.key_off ? rows.key_off[clip] : 0;       // see attn_dec_kernel
  const int n_keys = n_keys_base + (pos_dev ? *pos_dev : 0) + rows.key_step * (b % rows.group) - k_off;
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) float part_o[AD_WAVES][64];
  __shared__ float part_m[AD_WAVES], part_l[AD_WAVES];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x;
  if (n_keys <= 0) {
    if (tid < 64) out[(long)b * ldo + h * 64 + tid] = 0.f;
    return;
  }
  const int c = lane & 7, r = lane >> 3;
  const _Float16* Kb = kv + (long)clip * kv_batch_stride + koff + h * head_stride + 8 * c + (long)k_off * ldkv;
  const _Float16* Vb = kv + (long)clip * kv_batch_stride + voff + h * head_stride + 8 * c + (long)k_off * ldkv;
  const int per = (n_keys + AD_WAVES - 1) / AD_WAVES;
  const int k_lo = wave * per, k_hi = min(n_keys, k_lo + per);
  const int k_last = max(k_hi - 1, 0);          // clamp target of the slots past the partition (weight 0)
  half8 kr[ADX_SLOTS], vr[ADX_SLOTS];
  // rows.stream_kv: the K|V of a decode step over many clips is a one-pass stream (0.6 GB per position at 64 tiny clips):
  // requested non-temporally it no longer pushes the decoder's weights (3 MB per XCD, re-read every step) out of the
  // L2s: 11.25 -> 10.45 ms per 36 positions at 64 tiny clips, 46.4 -> 43.7 ms at 256 base clips.  For a few clips the
  // K|V itself is what stays cached from step to step, and the plain loads are 9 % faster (one clip: 5.95 vs 6.5 ms).
  // (a template parameter: a run-time choice between the two kinds of load -- select or branch -- is folded into one plain
  // load by the optimiser)
#pragma unroll
  for (int i = 0; i < ADX_SLOTS; ++i) {
    const half8* p = reinterpret_cast<const half8*>(Kb + (long)min(k_lo + 8 * i + r, k_last) * ldkv);
    kr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
  }
#pragma unroll
  for (int i = 0; i < ADX_SLOTS; ++i) {
    const half8* p = reinterpret_cast<const half8*>(Vb + (long)min(k_lo + 8 * i + r, k_last) * ldkv);
    vr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
  }
  __builtin_amdgcn_sched_barrier(0);            // the scheduler would otherwise keep 9 loads in flight and interleave the rest
  float qv[8];
  {
    const float4 q0 = *reinterpret_cast<const float4*>(q + (long)b * ldq + h * 64 + 8 * c);
    const float4 q1 = *reinterpret_cast<const float4*>(q + (long)b * ldq + h * 64 + 8 * c + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w;
    qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    if (rows.attn16) {       // the query as ggml's K.q sees it: rounded BEFORE the scaling (q / 8 may be an f16 subnormal where q is not)
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = (float)(_Float16)qv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
  }
  float sc[ADX_SLOTS];
  float mloc = -1e30f;
#pragma unroll
  for (int i = 0; i < ADX_SLOTS; ++i) {
    float v = (float)kr[i][0] * qv[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) v = fmaf((float)kr[i][e], qv[e], v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // lanes ^ 1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // lanes ^ 2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // the other quad of the 8
    const bool valid = k_lo + 8 * i + r < k_hi;
    sc[i] = valid ? v : -1e30f;
    mloc = fmaxf(mloc, sc[i]);
  }
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
  float lsum = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float inv16 = 0.f;           // attn16: 1 / (sum over ALL keys); the maximum is the row's, not the wave's
  if (rows.attn16) {
    if (lane == 0) part_m[wave] = mloc;
    __syncthreads();
    float m = part_m[0];
#pragma unroll
    for (int w = 1; w < AD_WAVES; ++w) m = fmaxf(m, part_m[w]);
    mloc = m;
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < ADX_SLOTS; ++i) ls += k_lo + 8 * i + r < k_hi ? __expf(sc[i] - mloc) : 0.f;
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) ls += __shfl_xor(ls, off, 64);     // the eight lanes of a key hold the same score
    if (lane == 0) part_l[wave] = ls;
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < AD_WAVES; ++w) l += part_l[w];
    inv16 = 1.f / l;
    __syncthreads();           // part_m / part_l are written again below
  }
#pragma unroll
  for (int i = 0; i < ADX_SLOTS; ++i) {
    float pw = k_lo + 8 * i + r < k_hi ? __expf(sc[i] - mloc) : 0.f;
    if (rows.attn16) pw = (float)(_Float16)(pw * inv16);      // the normalised probability as ggml's P.V sees it
    lsum += pw;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(pw, (float)vr[i][e], acc[e]);
  }
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) {
    lsum += __shfl_xor(lsum, off, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
  }
  if (r == 0) {
    *reinterpret_cast<float4*>(&part_o[wave][8 * c]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(&part_o[wave][8 * c + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  if (lane == 0) { part_m[wave] = mloc; part_l[wave] = lsum; }
  __syncthreads();
  if (wave == 0) {
    float m = part_m[0];
#pragma unroll
    for (int w = 1; w < AD_WAVES; ++w) m = fmaxf(m, part_m[w]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < AD_WAVES; ++w) {
      const float scl = __expf(part_m[w] - m);   // empty partitions have m = -1e30 -> scale 0
      o = fmaf(part_o[w][lane], scl, o);
      l = fmaf(part_l[w], scl, l);
    }
    // attn16: the probabilities were normalised before they were rounded (every wave's maximum is the row's: scl = 1)
    out[(long)b * ldo + h * 64 + lane] = rows.attn16 ? o : o / l;
  }
}

// token + positional embedding for one decode step: x[b][:] = tok_emb[token[b]] + pos_emb[pos]
__global__ __launch_bounds__(256) void embed_kernel(const int* __restrict__ tokens, const float* __restrict__ tok_emb,
                                                    const float* __restrict__ pos_emb, int pos,
                                                    const int* __restrict__ pos_dev, float* __restrict__ x, int D, int rpc,
                                                    const int* __restrict__ row_off) {
  const int b = blockIdx.x;
  const int tok = tokens[b];
  if (pos_dev) pos = *pos_dev;
  pos += b % rpc;                       // rpc rows per clip (the batched prompt step): consecutive positions
  if (row_off) pos = max(pos - row_off[b / rpc], 0);      // left-padded prompts: cache row -> position of this clip
  for (int c = threadIdx.x; c < D; c += 256) x[(long)b * D + c] = tok_emb[(long)tok * D + c] + pos_emb[(long)pos * D + c];
}

// greedy pick: argmax over the vocabulary with a suppression mask (mask[v] != 0 -> -inf); ties -> lowest id.
// One 1024-thread block per clip; every thread walks the row in float4 / uchar4 steps with four loads in flight
// (the 256-thread, one-float-per-iteration version ran at 0.1 TB/s: 118 us for 64 x 51865 logits).
constexpr int PICK_UNROLL = 13;                // float4 per thread of a 1024-thread pick: 53 248 logits in one round of requests
__device__ __forceinline__ unsigned load_u32_unaligned(const unsigned char* p) {
  unsigned v;
  __builtin_memcpy(&v, p, 4);                  // global memory takes unaligned dword loads; the compiler emits one
  return v;
}
// the tail of a fused pick: embedding of the pick for the next step, then the counters by the last workgroup to finish
__device__ __forceinline__ void step_fuse_embed(const StepFuse& f, int tok, int pos, int b, int tid) {
  const int pe = f.row_off ? pos - f.row_off[b] : pos;    // cache row `pos` is position pos - row_off[b] of a left-padded clip
  if (f.tok_emb_q) {
    for (int c = tid; c < f.D; c += (int)blockDim.x)
      f.x[(long)b * f.D + c] = q_elem(f.tok_emb_q, f.tok_emb_ttype, (long)tok * f.D + c) + f.pos_emb[(long)pe * f.D + c];
  } else {
    for (int c = tid; c < f.D; c += (int)blockDim.x) f.x[(long)b * f.D + c] = f.tok_emb[(long)tok * f.D + c] + f.pos_emb[(long)pe * f.D + c];
  }
}
__device__ __forceinline__ void step_fuse_ticket(const StepFuse& f, int pos, int step, int tid) {
  if (tid == 0) {
    // No fence: nothing of this kernel is read by another workgroup of it -- the ticket only elects the workgroup that
    // stores the counters, every workgroup has consumed the old values long before its own increment, and the
    // kernels behind read everything after the kernel boundary.  (A device-scope fence here writes back and
    // invalidates the XCD's L2 once per workgroup and step.)
    const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(f.counters + 2), 1u);
    if (t == gridDim.x - 1) {              // everyone else has finished, so everyone has read the old counters
      f.counters[0] = pos;
      f.counters[1] = step + 1;
      f.counters[2] = 0;
    }
  }
}
__device__ __forceinline__ void step_fuse_tail(const StepFuse& f, int tok, int pos, int step, int b, int tid) {
  step_fuse_embed(f, tok, pos, b, tid);
  step_fuse_ticket(f, pos, step, tid);
}

__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ mask,
                                                      const unsigned char* __restrict__ mask_first,
                                                      const int* __restrict__ step_dev, int V, long ld, int* __restrict__ tokens_out,
                                                      int* __restrict__ tokens_all, float* __restrict__ best_logit,
                                                      int eot, int* __restrict__ finished, int* __restrict__ done_count, StepFuse fuse) {
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int s_tok;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int step = fuse.x ? fuse.counters[1] : step_dev ? *step_dev : 0;
  const int pos = fuse.x ? fuse.counters[0] + 1 : 0;
  if (step == 0 && mask_first) mask = mask_first;
  const float* lg = logits + (long)b * ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  auto consider = [&](float x, int v) {
    if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
  };
  // rows start 4-byte aligned only (V is odd): scalar head up to the first 16-byte boundary, vector body, scalar tail
  const int head = (int)(((16 - ((size_t)lg & 15)) & 15) >> 2);
  if (tid < head && tid < V) consider((mask && mask[tid]) ? -INFINITY : lg[tid], tid);
  const int nvec = (V - head) >> 2;
  const float4* lg4 = reinterpret_cast<const float4*>(lg + head);
  // The whole row in flight at once: PICK_UNROLL x 1024 float4 cover 53 248 logits, requested from clamped addresses
  // before any is looked at (with `unroll 4` the scan was four dependent round trips), and the four mask bytes of a
  // float4 as ONE unaligned 32-bit load (rows and so v0 are only 4-byte aligned in the logits, 1-byte in the mask).
  {
    float4 xs[PICK_UNROLL];
    unsigned ms[PICK_UNROLL];
#pragma unroll
    for (int u = 0; u < PICK_UNROLL; ++u) {
      const int q = min(tid + 1024 * u, nvec - 1);
      xs[u] = lg4[q];
      ms[u] = mask ? load_u32_unaligned(mask + head + 4 * q) : 0u;
    }
#pragma unroll
    for (int u = 0; u < PICK_UNROLL; ++u) {
      const int q = tid + 1024 * u;
      if (q < nvec) {
        const int v0 = head + 4 * q;
        consider((ms[u] & 0xffu) ? -INFINITY : xs[u].x, v0);
        consider((ms[u] & 0xff00u) ? -INFINITY : xs[u].y, v0 + 1);
        consider((ms[u] & 0xff0000u) ? -INFINITY : xs[u].z, v0 + 2);
        consider((ms[u] & 0xff000000u) ? -INFINITY : xs[u].w, v0 + 3);
      }
    }
  }
  for (int q = tid + 1024 * PICK_UNROLL; q < nvec; q += 1024) {          // vocabularies beyond 53 248 entries
    const float4 x = lg4[q];
    const int v0 = head + 4 * q;
    unsigned char m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    if (mask) { m0 = mask[v0]; m1 = mask[v0 + 1]; m2 = mask[v0 + 2]; m3 = mask[v0 + 3]; }
    consider(m0 ? -INFINITY : x.x, v0);
    consider(m1 ? -INFINITY : x.y, v0 + 1);
    consider(m2 ? -INFINITY : x.z, v0 + 2);
    consider(m3 ? -INFINITY : x.w, v0 + 3);
  }
  for (int v = head + 4 * nvec + tid; v < V; v += 1024) consider((mask && mask[v]) ? -INFINITY : lg[v], v);
  // wave arg-max (ties -> lowest id), then the 16 wave results
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    tokens_out[b] = bi;
    if (tokens_all) tokens_all[(long)step * gridDim.x + b] = bi;
    if (best_logit) best_logit[(long)step * gridDim.x + b] = bv;
    // first EOT of this clip: the host polls done_count and stops replaying the step graph once every clip has one
    if (finished && bi == eot && !finished[b]) { finished[b] = 1; atomicAdd(done_count, 1); }
    s_tok = bi;
  }
  if (fuse.x) {
    __syncthreads();
    step_fuse_tail(fuse, s_tok, pos, step, b, tid);
  }
}

// What may be picked next in a window under the timestamp rules (oracle/whisper_oracle.py: _apply_rules), from the
// window's state; uniform per clip and step.
struct TsRule {
  const unsigned char* mask;
  int beg, not_tok, ts_lo, ts_hi;     // timestamps below ts_lo and above ts_hi are not allowed
  bool no_ts, forced_ts, text_allowed;
  __device__ __forceinline__ TsRule(const TsPickArgs& a, const TsState& st) {
    mask = (st.n == 0 && a.mask_first) ? a.mask_first : a.mask;
    beg = a.beg; not_tok = a.not_tok;
    const bool last_ts = st.n >= 1 && st.last >= a.beg;
    const bool pen_ts = st.n < 2 || st.prev >= a.beg;
    ts_lo = a.beg; ts_hi = a.V - 1;
    if (st.last_ts >= 0) ts_lo = (a.rules == TS_RULES_OPENAI && !(last_ts && !pen_ts)) ? st.last_ts + 1 : st.last_ts;
    const bool initial = st.n == 0;
    if (initial && a.max_initial_ts > 0) ts_hi = a.beg + a.max_initial_ts;
    no_ts = last_ts && pen_ts;
    forced_ts = initial && a.rules == TS_RULES_OPENAI;      // the first pick is a timestamp
    text_allowed = !(last_ts && !pen_ts) && !forced_ts;     // ids < eot
  }
  // ids >= eot (EOT, specials, timestamps)
  __device__ __forceinline__ bool masked_hi(int v) const {
    if (mask && mask[v]) return true;
    if (v == not_tok) return true;
    if (v >= beg) return no_ts || v < ts_lo || v > ts_hi;
    return forced_ts;
  }
  __device__ __forceinline__ bool allowed(int v, int eot) const {
    if (v < eot) return text_allowed && !(mask && mask[v]);
    return !masked_hi(v);
  }
  // the same with the id's suppression byte already loaded (the sampling pick reads eight at a time)
  __device__ __forceinline__ bool allowed_m(int v, int eot, bool masked) const {
    if (masked) return false;
    if (v < eot) return text_allowed;
    if (v == not_tok) return false;
    if (v >= beg) return !(no_ts || v < ts_lo || v > ts_hi);
    return !forced_ts;
  }
};

// record a pick and move the window's state on (thread 0 of the clip's workgroup)
__device__ __forceinline__ void ts_commit(const TsPickArgs& a, TsState& st, int b, int step, int pick, int tsid, float plog) {
  a.tokens_out[b] = pick;
  a.tokens_all[(long)step * gridDim.x + b] = pick;
  a.tids_all[(long)step * gridDim.x + b] = tsid;
  if (a.plog_all) a.plog_all[(long)step * gridDim.x + b] = plog;
  st.prev = st.last;
  st.last = pick;
  st.n += 1;
  if (a.rules == TS_RULES_OPENAI ? pick >= a.beg : pick > a.beg) st.last_ts = pick;
  bool done = pick == a.eot;
  if (a.rules == TS_RULES_WCPP && st.last_ts >= 0 && st.seek + 2 * (st.last_ts - a.beg) + a.delta_min >= st.seek_end) done = true;
  if (done) { st.done = 1; atomicAdd(a.done_count, 1); }
  a.st[b] = st;
}

// greedy pick under the timestamp rules: one 1024-thread block per clip.  Plain text tokens [0, eot) are either all
// subject to the suppression mask only or not allowed at all (uniform per clip and step), so they are scanned in
// float4 steps or skipped; the ~1600 special and timestamp ids take the per-id rule path.  The row stays in registers
// between the two passes: maxima (the pick, the reference point of the exponentials), then the two sums -- over
// everything allowed (the pick's log-probability, whisper_token_data::plog) and over the allowed timestamps (the
// probability-mass rule).
__global__ __launch_bounds__(1024) void ts_pick_kernel(TsPickArgs a) {
  __shared__ float s_v[2][16];
  __shared__ int s_i[2][16];
  __shared__ float s_sum[2][16];
  __shared__ int s_tok;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int step = a.fuse.x ? a.fuse.counters[1] : a.step_dev ? *a.step_dev : 0;
  const int pos = a.fuse.x ? a.fuse.counters[0] + 1 : 0;
  TsState st = a.st[b];
  if (st.done) {   // finished window: keep feeding EOT so that the batch stays in lock step
    if (tid == 0) {
      a.tokens_out[b] = a.eot;
      a.tokens_all[(long)step * gridDim.x + b] = a.eot;
      a.tids_all[(long)step * gridDim.x + b] = a.beg;
      if (a.plog_all) a.plog_all[(long)step * gridDim.x + b] = 0.f;
    }
    if (a.fuse.x) step_fuse_tail(a.fuse, a.eot, pos, step, b, tid);
    return;
  }
  const float* lg = a.logits + (long)b * a.ld;
  const TsRule rule(a, st);
  const unsigned char* mask = rule.mask;
  const bool text_allowed = rule.text_allowed;
  // the text range [0, eot): scalar head up to the first 16-byte boundary, float4 body in registers, scalar tail
  const int head = min((int)(((16 - ((size_t)lg & 15)) & 15) >> 2), a.eot);
  const int nvec = (a.eot - head) >> 2;
  const float4* lg4 = reinterpret_cast<const float4*>(lg + head);
  float4 xs[PICK_UNROLL];
  if (text_allowed) {                          // the whole text range in flight at once (see argmax_kernel)
    unsigned ms[PICK_UNROLL];
#pragma unroll
    for (int u = 0; u < PICK_UNROLL; ++u) {
      const int q = min(tid + 1024 * u, nvec - 1);
      xs[u] = lg4[q];
      ms[u] = mask ? load_u32_unaligned(mask + head + 4 * q) : 0u;
    }
#pragma unroll
    for (int u = 0; u < PICK_UNROLL; ++u) {
      if (ms[u] & 0xffu) xs[u].x = -INFINITY;
      if (ms[u] & 0xff00u) xs[u].y = -INFINITY;
      if (ms[u] & 0xff0000u) xs[u].z = -INFINITY;
      if (ms[u] & 0xff000000u) xs[u].w = -INFINITY;
    }
  }
  // the ids from EOT up (EOT, specials, timestamps: 1 608 of them, two per thread): value and rule evaluated once, kept in
  // registers for both passes (re-loading them in the second pass was another exposed round trip to the logits)
  float hx[2];
  bool hok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = a.eot + tid + 1024 * k;
    hok[k] = v < a.V && !rule.masked_hi(min(v, a.V - 1));
    hx[k] = lg[min(v, a.V - 1)];
  }
  // every allowed (value, id) of this thread's share of the row: the plain text ids through FT(x, v), the ids from EOT up
  // through FH(x, v).  (A macro, not a lambda: closures that capture the running maxima by reference ended up in scratch.)
#define TS_SCAN(FT, FH)                                                                                              \
  do {                                                                                                               \
    if (text_allowed) {                                                                                              \
      if (tid < head) { const float x_ = (mask && mask[tid]) ? -INFINITY : lg[tid]; FT(x_, tid); }                   \
      _Pragma("unroll") for (int u = 0; u < PICK_UNROLL; ++u) {                                                      \
        const int q = tid + 1024 * u;                                                                                \
        if (q < nvec) {                                                                                              \
          const int v0 = head + 4 * q;                                                                               \
          FT(xs[u].x, v0); FT(xs[u].y, (v0 + 1)); FT(xs[u].z, (v0 + 2)); FT(xs[u].w, (v0 + 3));                      \
        }                                                                                                            \
      }                                                                                                              \
      for (int q = tid + 1024 * PICK_UNROLL; q < nvec; q += 1024) { /* vocabularies beyond 53 248 entries */         \
        float4 x = lg4[q];                                                                                           \
        const int v0 = head + 4 * q;                                                                                 \
        if (mask) {                                                                                                  \
          if (mask[v0]) x.x = -INFINITY;                                                                             \
          if (mask[v0 + 1]) x.y = -INFINITY;                                                                         \
          if (mask[v0 + 2]) x.z = -INFINITY;                                                                         \
          if (mask[v0 + 3]) x.w = -INFINITY;                                                                         \
        }                                                                                                            \
        FT(x.x, v0); FT(x.y, (v0 + 1)); FT(x.z, (v0 + 2)); FT(x.w, (v0 + 3));                                        \
      }                                                                                                              \
      for (int v = head + 4 * nvec + tid; v < a.eot; v += 1024) {                                                    \
        const float x_ = (mask && mask[v]) ? -INFINITY : lg[v];                                                      \
        FT(x_, v);                                                                                                   \
      }                                                                                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                                    \
      if (hok[k]) FH(hx[k], (a.eot + tid + 1024 * k));                                                               \
    for (int v = a.eot + tid + 2048; v < a.V; v += 1024)      /* more than 2 048 ids from EOT up: no such vocabulary */ \
      if (!rule.masked_hi(v)) { const float x_ = lg[v]; FH(x_, v); }                                                 \
  } while (0)
  float tv = -INFINITY, xv = -INFINITY;   // best text (v < beg) and best timestamp
  int ti = 0x7fffffff, xi = 0x7fffffff;
#define TS_MAX_T(x, v) do { if ((x) > tv || ((x) == tv && (v) < ti)) { tv = (x); ti = (v); } } while (0)
#define TS_MAX_H(x, v) do { if ((v) < a.beg) TS_MAX_T(x, v); else if ((x) > xv || ((x) == xv && (v) < xi)) { xv = (x); xi = (v); } } while (0)
  TS_SCAN(TS_MAX_T, TS_MAX_H);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float ov = __shfl_xor(tv, off, 64); int oi = __shfl_xor(ti, off, 64);
    if (ov > tv || (ov == tv && oi < ti)) { tv = ov; ti = oi; }
    ov = __shfl_xor(xv, off, 64); oi = __shfl_xor(xi, off, 64);
    if (ov > xv || (ov == xv && oi < xi)) { xv = ov; xi = oi; }
  }
  if ((tid & 63) == 0) { s_v[0][tid >> 6] = tv; s_i[0][tid >> 6] = ti; s_v[1][tid >> 6] = xv; s_i[1][tid >> 6] = xi; }
  __syncthreads();
  float max_text = s_v[0][0], max_ts = s_v[1][0];
  int arg_text = s_i[0][0], arg_ts = s_i[1][0];
#pragma unroll
  for (int w = 1; w < 16; ++w) {
    if (s_v[0][w] > max_text || (s_v[0][w] == max_text && s_i[0][w] < arg_text)) { max_text = s_v[0][w]; arg_text = s_i[0][w]; }
    if (s_v[1][w] > max_ts || (s_v[1][w] == max_ts && s_i[1][w] < arg_ts)) { max_ts = s_v[1][w]; arg_ts = s_i[1][w]; }
  }
  // sums of exponentials: everything allowed against the overall maximum, the allowed timestamps against theirs
  const float max_all = fmaxf(max_text, max_ts);
  float sum_all = 0.f, sum = 0.f;
#define TS_SUM_T(x, v) do { (void)(v); sum_all += __expf((x) - max_all); } while (0)      /* (a masked value is -inf: adds 0) */
#define TS_SUM_H(x, v) do { sum_all += __expf((x) - max_all); if ((v) >= a.beg) sum += expf((x) - max_ts); } while (0)
  if (max_all > -INFINITY) TS_SCAN(TS_SUM_T, TS_SUM_H);
#undef TS_SUM_T
#undef TS_SUM_H
#undef TS_MAX_T
#undef TS_MAX_H
#undef TS_SCAN
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); sum_all += __shfl_xor(sum_all, off, 64); }
  if ((tid & 63) == 0) { s_sum[0][tid >> 6] = sum; s_sum[1][tid >> 6] = sum_all; }
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f, tot_all = 0.f;
    for (int w = 0; w < 16; ++w) { tot += s_sum[0][w]; tot_all += s_sum[1][w]; }
    const float lse_ts = max_ts > -INFINITY ? max_ts + logf(tot) : -INFINITY;
    int pick;
    if (lse_ts > max_text) pick = arg_ts;                       // timestamps carry more mass than any text token
    else pick = (max_ts > max_text) ? arg_ts : arg_text;        // plain arg-max, ties -> lowest id (text ids are lower)
    if (pick == 0x7fffffff) pick = a.eot;                       // everything masked: cannot happen with sane masks
    const int tsid = pick >= a.beg ? pick : (max_ts > -INFINITY ? arg_ts : a.beg);
    const float plog = (pick >= a.beg ? max_ts : max_text) - (max_all + logf(tot_all));
    ts_commit(a, st, b, step, pick, tsid, plog);
    s_tok = pick;
  }
  if (a.fuse.x) {
    __syncthreads();
    step_fuse_tail(a.fuse, s_tok, pos, step, b, tid);
  }
}

// The sampling form of the pick (whisper_sample_token(best = false) at a temperature > 0 [UPSTREAM-RECALL]): logits /
// temperature, the same rules, probabilities = exp(log-softmax) with the text tokens removed when the probability-mass
// rule fires, then std::discrete_distribution -- the first id whose cumulative share of the (re-normalised) probability
// reaches the uniform variate u the host drew for this (step, row).  Only the fallback path of whisper_full comes here
// (a window the greedy pass failed on), so the kernel is plain: thread t owns the ids [t * chunk, (t + 1) * chunk),
// three passes over them (maxima, sums, probabilities), cumulative sums in double.
// The sampling pick: 1024 threads per row.  Ids are dealt out in blocks of 8192: thread t owns the eight consecutive ids
// 8192 j + 8 t .. + 7 of every block j (7 blocks cover n_vocab <= 57 344: the launcher checks), so a wave's loads are two
// contiguous float4 runs per block -- and one 8-byte run of the suppression mask -- instead of 64 scattered lines per id
// (the first form, one contiguous range of ids per thread: 148 us per pick, most of a call that falls back through the
// temperature ladder).  Three passes over the row (maximum; sums; probabilities), each re-reading it from the L2 it was
// just written to: keeping the row in registers across the passes was tried in three shapes and spilled in all of them
// (56 - 104 values per lane beside three inlined expf per value).  The cumulative distribution runs in id order =
// (block, thread, element): per block an inclusive scan of the threads' eight-id sums in double, block bases from a
// 7 x 16 table of wave totals; the thread whose run contains u walks its eight ids.
constexpr int TS_THREADS = 1024, TS_WAVES = TS_THREADS / 64, TS_E = 8, TS_BLK = TS_THREADS * TS_E, TS_NB = 7;
static_assert(TS_SCRATCH_ROW == TS_NB * TS_BLK, "asr_common.h sizes the sampling pick's row scratch");
// x[e] = logit / T where the rules allow id v0 + e, -inf where they do not or past the row
__device__ __forceinline__ void ts_load_run(const TsRule& rule, const float* __restrict__ lg, int V, int eot, float T, int v0, float (&x)[TS_E]) {
  float raw[TS_E];
  unsigned char mk[TS_E];
  if (v0 + TS_E <= V) {                 // whole run inside the row: rows are 16-byte aligned (ld % 4 == 0), so is v0
    const float4 q0 = *reinterpret_cast<const float4*>(lg + v0), q1 = *reinterpret_cast<const float4*>(lg + v0 + 4);
    raw[0] = q0.x; raw[1] = q0.y; raw[2] = q0.z; raw[3] = q0.w; raw[4] = q1.x; raw[5] = q1.y; raw[6] = q1.z; raw[7] = q1.w;
    uint2 m8 = make_uint2(0u, 0u);
    if (rule.mask) m8 = *reinterpret_cast<const uint2*>(rule.mask + v0);
#pragma unroll
    for (int e = 0; e < TS_E; ++e) mk[e] = (unsigned char)((e < 4 ? m8.x >> (8 * e) : m8.y >> (8 * (e - 4))) & 0xffu);
  } else {
#pragma unroll
    for (int e = 0; e < TS_E; ++e) {
      const int v = v0 + e;
      raw[e] = v < V ? lg[v] : 0.f;
      mk[e] = (v < V && rule.mask) ? rule.mask[v] : 0;
    }
  }
#pragma unroll
  for (int e = 0; e < TS_E; ++e) {
    const int v = v0 + e;
    x[e] = (v < V && rule.allowed_m(v, eot, mk[e] != 0)) ? raw[e] / T : -INFINITY;
  }
}
__device__ __forceinline__ void ts_read_run(const float* __restrict__ xrow, int v0, float (&x)[TS_E]) {     // the thread's own stores
  const float4 q0 = *reinterpret_cast<const float4*>(xrow + v0), q1 = *reinterpret_cast<const float4*>(xrow + v0 + 4);
  x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
}
__global__ __launch_bounds__(TS_THREADS) void ts_sample_kernel(TsPickArgs a) {
  __shared__ float s_v[2][TS_WAVES];
  __shared__ int s_i[TS_WAVES];
  __shared__ float s_sum[2][TS_WAVES];
  __shared__ double s_wtot[TS_NB][TS_WAVES];
  __shared__ double s_own[TS_NB][TS_THREADS], s_pre[TS_NB][TS_THREADS];     // a run's sum / what its wave holds in front of it (2 x 56 KB)
  __shared__ int s_tok, s_last;
  __shared__ float s_px;
  __shared__ int s_ctok[TS_MAX_CAND];
  __shared__ float s_cpx[TS_MAX_CAND];
  __shared__ double s_cu[TS_MAX_CAND];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int step = a.fuse.x ? a.fuse.counters[1] : a.step_dev ? *a.step_dev : 0;
  const int pos = a.fuse.x ? a.fuse.counters[0] + 1 : 0;
  TsState st = a.st[b];
  const int n_cand = a.n_cand;
  if (st.done) {
    if (n_cand > 0) return;             // beam search: a finished decoder draws nothing
    if (tid == 0) {
      a.tokens_out[b] = a.eot;
      a.tokens_all[(long)step * gridDim.x + b] = a.eot;
      a.tids_all[(long)step * gridDim.x + b] = a.beg;
      if (a.plog_all) a.plog_all[(long)step * gridDim.x + b] = 0.f;
    }
    if (a.fuse.x) step_fuse_tail(a.fuse, a.eot, pos, step, b, tid);
    return;
  }
  const float* lg = a.logits + (long)b * a.ld;
  const TsRule rule(a, st);
  const float T = *a.temperature;
  const double u = n_cand > 0 ? 0.0 : a.u_all[(long)step * gridDim.x + b];
  if (tid == 0) { s_tok = -1; s_last = -1; }
  if (tid < n_cand) { s_ctok[tid] = -1; s_cu[tid] = a.u_all[((long)step * gridDim.x + b) * n_cand + tid]; }
  // ---- pass 1: maxima of the text / special ids and of the timestamps ----
  float tv = -INFINITY, xv = -INFINITY;
  int xi = 0x7fffffff;
  // The row as the passes see it (logit / T where the rules allow the id, -inf elsewhere) is evaluated ONCE and parked in
  // xrow (global, this row's TS_NB x TS_BLK floats): passes 2 and 3 read a thread's own eight-id runs back instead of
  // re-deriving them -- the rules and the IEEE division were 44 of the ~126 instructions per id of the three passes.
  float* __restrict__ xrow = a.x_scratch + (long)b * (TS_NB * TS_BLK);
#pragma unroll 1
  for (int jb = 0; jb < TS_NB; ++jb) {
    const int v0 = TS_BLK * jb + TS_E * tid;
    float x[TS_E];
    ts_load_run(rule, lg, a.V, a.eot, T, v0, x);
    *reinterpret_cast<float4*>(xrow + v0) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<float4*>(xrow + v0 + 4) = make_float4(x[4], x[5], x[6], x[7]);
#pragma unroll
    for (int e = 0; e < TS_E; ++e) {
      if (v0 + e < a.beg) tv = fmaxf(tv, x[e]);
      else if (x[e] > xv) { xv = x[e]; xi = v0 + e; }          // a thread's ids ascend: its first maximum stays
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    tv = fmaxf(tv, __shfl_xor(tv, off, 64));
    const float ov = __shfl_xor(xv, off, 64); const int oi = __shfl_xor(xi, off, 64);
    if (ov > xv || (ov == xv && oi < xi)) { xv = ov; xi = oi; }
  }
  if (lane == 0) { s_v[0][wave] = tv; s_v[1][wave] = xv; s_i[wave] = xi; }
  __syncthreads();
  float max_text = s_v[0][0], max_ts = s_v[1][0];
  int arg_ts = s_i[0];
#pragma unroll
  for (int w = 1; w < TS_WAVES; ++w) {
    max_text = fmaxf(max_text, s_v[0][w]);
    if (s_v[1][w] > max_ts || (s_v[1][w] == max_ts && s_i[w] < arg_ts)) { max_ts = s_v[1][w]; arg_ts = s_i[w]; }
  }
  const float max_all = fmaxf(max_text, max_ts);
  // ---- pass 2: sums of exponentials ----
  float sum_all = 0.f, sum_ts = 0.f;
#pragma unroll 1
  for (int jb = 0; jb < TS_NB; ++jb) {
    const int v0 = TS_BLK * jb + TS_E * tid;
    float x[TS_E];
    ts_read_run(xrow, v0, x);
#pragma unroll
    for (int e = 0; e < TS_E; ++e) {
      if (x[e] == -INFINITY) continue;                         // not allowed, or past the row
      sum_all += expf(x[e] - max_all);
      if (v0 + e >= a.beg) sum_ts += expf(x[e] - max_ts);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sum_all += __shfl_xor(sum_all, off, 64); sum_ts += __shfl_xor(sum_ts, off, 64); }
  if (lane == 0) { s_sum[0][wave] = sum_all; s_sum[1][wave] = sum_ts; }
  __syncthreads();
  float tot_all = 0.f, tot_ts = 0.f;
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) { tot_all += s_sum[0][w]; tot_ts += s_sum[1][w]; }
  const float lse_all = max_all + logf(tot_all);
  const float lse_ts = max_ts > -INFINITY ? max_ts + logf(tot_ts) : -INFINITY;
  const bool ts_only = lse_ts > max_text;                      // the probability-mass rule
  // ---- pass 3: probabilities, eight-id sums per block, and per block an inclusive scan over the wave ----
  int last_pos = -1;
#pragma unroll 1
  for (int jb = 0; jb < TS_NB; ++jb) {
    const int v0 = TS_BLK * jb + TS_E * tid;
    float x[TS_E];
    ts_read_run(xrow, v0, x);
    double sj = 0.0;
#pragma unroll
    for (int e = 0; e < TS_E; ++e) {
      float p = 0.f;
      if (x[e] != -INFINITY && !(ts_only && v0 + e < a.beg)) p = expf(x[e] - lse_all);
      if (p > 0.f) { sj += (double)p; last_pos = v0 + e; }
    }
    double in = sj;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(in, off, 64);
      if (lane >= off) in += o;
    }
    s_own[jb][tid] = sj;
    s_pre[jb][tid] = in - sj;
    if (lane == 63) s_wtot[jb][wave] = in;
  }
  if (last_pos >= 0) atomicMax(&s_last, last_pos);
  __syncthreads();
  double total = 0.0;
#pragma unroll
  for (int jb = 0; jb < TS_NB; ++jb)
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) total += s_wtot[jb][w];
  // the run whose interval (cum before, cum after] / total contains u is walked by its thread; the last id with any
  // probability closes at 1.0.  id order = (block, wave, lane, element).
  const int glast = s_last;
  double run = 0.0;                  // everything of the blocks before jb
#pragma unroll 1
  for (int jb = 0; jb < TS_NB; ++jb) {
    double front = run;              // + the waves of this block before mine
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) {
      if (w < wave) front += s_wtot[jb][w];
      run += s_wtot[jb][w];
    }
    const double own = s_own[jb][tid];
    if (!(own > 0.0)) continue;
    const double before = front + s_pre[jb][tid], after = before + own;
    const int v0 = TS_BLK * jb + TS_E * tid;
    const bool owns_last = glast >= v0 && glast < v0 + TS_E;
    const double lo = before / total, hi = owns_last ? 1.0 : after / total;
    // the walk of this run for one variate
    auto walk = [&](double uu, int* tok_out, float* px_out) {
      double c = before;
      int pick = -1;
      float px = 0.f;
      for (int e = 0; e < TS_E; ++e) {
        const int v = v0 + e;
        if (pick >= 0 || v >= a.V || (ts_only && v < a.beg) || !rule.allowed(v, a.eot)) continue;
        const float x = lg[v] / T;
        const float p = expf(x - lse_all);
        if (!(p > 0.f)) continue;
        c += (double)p;
        if (c / total >= uu || v == glast) { pick = v; px = x; }
      }
      if (pick < 0) { pick = glast; px = lg[glast] / T; }      // (rounding: the run's last id)
      *tok_out = pick;
      *px_out = px;
    };
    if (n_cand > 0) {
      for (int k = 0; k < n_cand; ++k) {
        const double uk = s_cu[k];
        if (lo < uk && uk <= hi) walk(uk, &s_ctok[k], &s_cpx[k]);
      }
    } else if (lo < u && u <= hi) {
      walk(u, &s_tok, &s_px);
    }
  }
  __syncthreads();
  if (n_cand > 0) {
    if (tid < n_cand) {
      int pick = s_ctok[tid];
      float px = s_cpx[tid];
      if (pick < 0) {                   // the variate fell between two runs' rounded interval ends: the last id with any probability
        pick = s_last >= 0 ? s_last : a.eot;
        px = lg[pick] / T;
      }
      a.cand_tok[(long)b * n_cand + tid] = pick;
      a.cand_plog[(long)b * n_cand + tid] = px - lse_all;
      a.cand_tid[(long)b * n_cand + tid] = pick >= a.beg ? pick : (max_ts > -INFINITY ? arg_ts : a.beg);
    }
    return;
  }
  if (tid == 0) {
    int pick = s_tok;
    float px = s_px;
    if (pick < 0) {                     // u fell between two runs' rounded interval ends: the last id with any probability
      pick = s_last >= 0 ? s_last : a.eot;
      px = lg[pick] / T;
    }
    const int tsid = pick >= a.beg ? pick : (max_ts > -INFINITY ? arg_ts : a.beg);
    ts_commit(a, st, b, step, pick, tsid, px - lse_all);
    s_tok = pick;
  }
  if (a.fuse.x) {
    __syncthreads();
    step_fuse_tail(a.fuse, s_tok, pos, step, b, tid);
  }
}

// ---------------------------------------------------------------------------------------------
// The same cross-attention for the query rows of ONE clip together (rows.group > 1, every row over the same keys:
// the positions of a batched prompt step, the decoders of a fallback pass): one workgroup per (head, clip, chunk of
// ADG_ROWS rows) holds the clip's K and V of the head in registers once and takes its rows one after the other -- with
// a workgroup per row the clip's K | V reaches the CUs once per row (from the XCD's L2 after the first), and a prompt
// step over 64 clips x 4 positions was 85 us per layer against 30 for the bytes from HBM.  Per row exactly
// attn_dec_x16_kernel's arithmetic (partition, per-wave triples, merge), so a row's bits do not depend on the form.
// The row loop is kept rolled and the f16 -> f32 conversions pinned inside it (hoisted, they are 96 more registers).
// ---------------------------------------------------------------------------------------------
constexpr int ADG_ROWS = 8;
template <bool STREAM_KV>
__global__ __launch_bounds__(64 * AD_WAVES) void attn_dec_x16g_kernel(const float* __restrict__ q, long ldq, const _Float16* __restrict__ kv,
                                                                      long kv_batch_stride, long ldkv, long head_stride, long koff, long voff,
                                                                      int n_keys, float* __restrict__ out, long ldo, AttnRows rows) {
  constexpr int SL = 12;
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) float part_o[ADG_ROWS][AD_WAVES][64];
  __shared__ float part_m[ADG_ROWS][AD_WAVES], part_l[ADG_ROWS][AD_WAVES];
  __shared__ float wave_m[ADG_ROWS][AD_WAVES], wave_l[ADG_ROWS][AD_WAVES];
  __shared__ float sc_s[ADG_ROWS][AD_WAVES][SL][8];          // the scores: 6 KB per row (in registers they do not fit beside the keys)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, clip = blockIdx.y;
  const int r_lo = blockIdx.z * ADG_ROWS, n_r = min(ADG_ROWS, rows.group - r_lo);
  const int c = lane & 7, r = lane >> 3;
  const _Float16* Kb = kv + (long)clip * kv_batch_stride + koff + h * head_stride + 8 * c;
  const _Float16* Vb = kv + (long)clip * kv_batch_stride + voff + h * head_stride + 8 * c;
  const int per = (n_keys + AD_WAVES - 1) / AD_WAVES;
  const int k_lo = wave * per, k_hi = min(n_keys, k_lo + per);
  const int k_last = max(k_hi - 1, 0);
  half8 kr[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const half8* p = reinterpret_cast<const half8*>(Kb + (long)min(k_lo + 8 * i + r, k_last) * ldkv);
    kr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
  }
  __builtin_amdgcn_sched_barrier(0);
  // scores of every row of the chunk from the keys in registers
#pragma unroll 1
  for (int j = 0; j < n_r; ++j) {
    const long b = (long)clip * rows.group + r_lo + j;
    float qv[8];
    const float4 q0 = *reinterpret_cast<const float4*>(q + b * ldq + h * 64 + 8 * c);
    const float4 q1 = *reinterpret_cast<const float4*>(q + b * ldq + h * 64 + 8 * c + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w;
    qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    if (rows.attn16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = (float)(_Float16)qv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
    float mloc = -1e30f;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      asm volatile("" : "+v"(kr[i]));
      float v = (float)kr[i][0] * qv[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) v = fmaf((float)kr[i][e], qv[e], v);
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));
      const float sc = k_lo + 8 * i + r < k_hi ? v : -1e30f;
      if (c == 0) sc_s[j][wave][i][r] = sc;
      mloc = fmaxf(mloc, sc);
    }
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
    if (lane == 0) wave_m[j][wave] = mloc;
  }
  // the values, into the registers the keys leave
  __builtin_amdgcn_sched_barrier(0);
  half8 vr[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const half8* p = reinterpret_cast<const half8*>(Vb + (long)min(k_lo + 8 * i + r, k_last) * ldkv);
    vr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  if (rows.attn16) {             // mode 2: every row's maximum and sum over ALL keys before a probability is rounded
#pragma unroll 1
    for (int j = 0; j < n_r; ++j) {
      float m = wave_m[j][0];
#pragma unroll
      for (int w = 1; w < AD_WAVES; ++w) m = fmaxf(m, wave_m[j][w]);
      float ls = 0.f;
#pragma unroll
      for (int i = 0; i < SL; ++i) ls += k_lo + 8 * i + r < k_hi ? __expf(sc_s[j][wave][i][r] - m) : 0.f;
#pragma unroll
      for (int off = 8; off <= 32; off <<= 1) ls += __shfl_xor(ls, off, 64);
      if (lane == 0) wave_l[j][wave] = ls;
    }
    __syncthreads();
  }
#pragma unroll 1
  for (int j = 0; j < n_r; ++j) {
    float mloc = wave_m[j][wave], inv16 = 0.f;
    if (rows.attn16) {
      float l = 0.f;
      mloc = wave_m[j][0];
#pragma unroll
      for (int w = 1; w < AD_WAVES; ++w) mloc = fmaxf(mloc, wave_m[j][w]);
#pragma unroll
      for (int w = 0; w < AD_WAVES; ++w) l += wave_l[j][w];
      inv16 = 1.f / l;
    }
    float lsum = 0.f;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      float pw = k_lo + 8 * i + r < k_hi ? __expf(sc_s[j][wave][i][r] - mloc) : 0.f;
      if (rows.attn16) pw = (float)(_Float16)(pw * inv16);
      lsum += pw;
      asm volatile("" : "+v"(vr[i]));
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pw, (float)vr[i][e], acc[e]);
    }
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) {
      lsum += __shfl_xor(lsum, off, 64);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
    }
    if (r == 0) {
      *reinterpret_cast<float4*>(&part_o[j][wave][8 * c]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(&part_o[j][wave][8 * c + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (lane == 0) { part_m[j][wave] = mloc; part_l[j][wave] = lsum; }
  }
  __syncthreads();
  if (wave < n_r) {              // row `wave` of the chunk: the merge of its 16 triples
    float m = part_m[wave][0];
#pragma unroll
    for (int w = 1; w < AD_WAVES; ++w) m = fmaxf(m, part_m[wave][w]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < AD_WAVES; ++w) {
      const float scl = __expf(part_m[wave][w] - m);
      o = fmaf(part_o[wave][w][lane], scl, o);
      l = fmaf(part_l[wave][w], scl, l);
    }
    out[((long)clip * rows.group + r_lo + wave) * ldo + h * 64 + lane] = rows.attn16 ? o : o / l;
  }
}

// beam_kv_reorder: phase 0 gathers the parents' bytes into scratch, phase 1 scatters them into the rows.  The bytes are the
// cache rows of the positions generated so far, [counters[4], counters[0]) -- read on the device: the launch is part of a
// captured step; counters[5] = max_new of the pass, the scratch rows' pitch in positions.
__global__ __launch_bounds__(256) void beam_kv_copy_kernel(char* __restrict__ kv, char* __restrict__ scratch, const int* __restrict__ parent,
                                                           int rows, long row_bytes, long pos_bytes, const int* __restrict__ counters,
                                                           int phase) {
  const int r = blockIdx.x, l = blockIdx.y;
  const int p = parent[r];
  if (p == r) return;
  const int pos0 = counters[4];
  const long len = (long)(counters[0] - pos0) * pos_bytes, scratch_stride = (long)counters[5] * pos_bytes;
  char* row = kv + ((long)l * rows + (phase == 0 ? p : r)) * row_bytes + (long)pos0 * pos_bytes;
  char* tmp = scratch + ((long)l * rows + r) * scratch_stride;
  for (long x = 16L * (blockIdx.z * 256 + threadIdx.x); x < len; x += 16L * 256 * gridDim.z) {
    if (phase == 0) *reinterpret_cast<uint4*>(tmp + x) = *reinterpret_cast<const uint4*>(row + x);
    else *reinterpret_cast<uint4*>(row + x) = *reinterpret_cast<const uint4*>(tmp + x);
  }
}

// One step of whisper_full's beam search for one clip (BeamArgs, asr_common.h): one wave, lane = candidate (decoder j, draw k).
__global__ __launch_bounds__(64) void beam_advance_kernel(BeamArgs a) {
  __shared__ BeamRow s_row[TS_MAX_CAND];
  __shared__ TsState s_st[TS_MAX_CAND];
  __shared__ double s_sum[64];
  __shared__ int s_tok[64], s_tid[64], s_sorted[64], s_deal[TS_MAX_CAND], s_feed[TS_MAX_CAND];
  __shared__ float s_plog[64];
  const int c = blockIdx.x, lane = threadIdx.x;
  const int n_dec = a.n_dec, n_cand = a.n_cand, r0 = c * n_dec;
  const int step = a.counters[1], pos = a.counters[0] + 1;
  if (lane < n_dec) { s_row[lane] = a.row[r0 + lane]; s_st[lane] = a.st[r0 + lane]; }
  __syncthreads();
  // the candidates of the clip's live decoders and the sum of ALL log-probabilities each would have
  const int j = lane / n_cand, k = lane - j * n_cand;
  const bool valid = j < n_dec && !(s_row[j < n_dec ? j : 0].completed || s_row[j < n_dec ? j : 0].failed);
  int tok = -1, tid = 0;
  float plog = 0.f;
  double sum = 0.0;
  if (valid) {
    const long x = (long)(r0 + j) * n_cand + k;
    tok = a.cand_tok[x]; tid = a.cand_tid[x]; plog = a.cand_plog[x];
    sum = s_row[j].sum_all + (double)plog;
  }
  s_tok[lane] = tok; s_tid[lane] = tid; s_plog[lane] = plog; s_sum[lane] = sum;
  const unsigned long long vmask = __ballot(valid);
  const int n_valid = __popcll(vmask);
  // std::stable_sort by (sum descending, decoder ascending) over candidates pushed decoder-major: position = the number of
  // candidates that come first = those with a larger sum, or an equal one and a smaller lane
  int rank = 0;
  for (int l = 0; l < 64; ++l) {
    const double sl = __shfl(sum, l, 64);
    if (((vmask >> l) & 1ull) && (sl > sum || (sl == sum && l < lane))) ++rank;
  }
  if (valid) s_sorted[rank] = lane;
  __syncthreads();
  // deal the sorted candidates to the live decoders, skipping repeats of the sequence just dealt (not at the first step):
  // two candidates are the same sequence when their ids are equal and their decoders' sequences are (BeamRow::eqid)
  if (lane == 0) {
    int cur_c = 0;
    for (int d = 0; d < n_dec; ++d) {
      s_deal[d] = -1;
      if (s_row[d].completed || s_row[d].failed || n_valid == 0) continue;
      if (cur_c >= n_valid) cur_c = 0;
      const int cur = s_sorted[cur_c++];
      const int cj = cur / n_cand;
      while (cur_c < n_valid && step > 0) {
        const int nx = s_sorted[cur_c], nj = nx / n_cand;
        if (!(s_tok[nx] == s_tok[cur] && (nj == cj || s_row[nj].eqid == s_row[cj].eqid))) break;
        ++cur_c;
      }
      s_deal[d] = cur;
    }
  }
  __syncthreads();
  int died = 0;
  if (lane < n_dec) {
    const int r = r0 + lane, cur = s_deal[lane];
    int feed = a.eot;
    if (cur < 0) {                               // ended in an earlier step: keeps its sequence and its cache rows
      a.parent[r] = r;
    } else {
      const int pj = cur / n_cand, t = s_tok[cur];
      BeamRow q = s_row[pj];
      TsState st = s_st[pj];
      q.sum_all = s_sum[cur];
      q.n += 1;
      st.prev = st.last; st.last = t; st.n += 1;                   // the rules' view of the sequence (the pick kernel's ts_commit)
      if (a.rules == TS_RULES_OPENAI ? t >= a.beg : t > a.beg) st.last_ts = t;
      // the sequence's class among the decoders dealt this step: the first decoder dealt the same parent class and id
      int eq = lane;
      for (int d = 0; d < lane; ++d) {
        const int od = s_deal[d];
        if (od >= 0 && s_tok[od] == t && s_row[od / n_cand].eqid == s_row[pj].eqid) { eq = d; break; }
      }
      q.eqid = eq;
      const long x = (long)step * a.rows + r;
      a.rec_tok[x] = t; a.rec_tid[x] = s_tid[cur]; a.rec_plog[x] = s_plog[cur]; a.rec_parent[x] = r0 + pj;
      a.parent[r] = r0 + pj;
      // completion / failure on the new last token (whisper_full's bookkeeping of a decoder)
      bool live = true;
      if (t > a.beg) {
        const int sd = 2 * (t - a.beg);
        if (q.has_ts && q.seek_delta > sd && q.result_len < step) { q.failed = 1; live = false; }      // "do not allow to go back in time"
        else { q.seek_delta = sd; q.result_len = step + 1; q.has_ts = 1; }
      }
      if (live && (t == a.eot || (q.has_ts && st.seek + q.seek_delta + a.delta_min >= st.seek_end))) {
        if (q.result_len == 0) {
          if (st.seek + q.seek_delta + a.delta_min >= st.seek_end) q.result_len = step + 1;
          else { q.failed = 1; live = false; }
        }
        if (live) { q.completed = 1; live = false; }
      }
      if (live && step == a.counters[5] - 1 && (q.result_len == 0 || q.seek_delta < 1500)) { q.failed = 1; live = false; }
      st.done = live ? 0 : 1;
      if (live) feed = t; else died = 1;
      a.row[r] = q;
      a.st[r] = st;
    }
    a.feed[r] = feed;
    s_feed[lane] = feed;
  }
  const int n_died = __popcll(__ballot(died != 0));
  if (lane == 0 && n_died) atomicAdd(a.done_count, n_died);
  if (!a.fuse.x) return;
  __syncthreads();
  for (int d = 0; d < n_dec; ++d) step_fuse_embed(a.fuse, s_feed[d], pos, r0 + d, lane);
  step_fuse_ticket(a.fuse, pos, step, lane);
}

// p_out[b] = softmax(row b)[token]: one 1024-thread block per row, two passes (maximum, sum of exponentials)
__global__ __launch_bounds__(1024) void softmax_prob_kernel(const float* __restrict__ logits, int V, long ld, int token, float* __restrict__ p_out) {
  __shared__ float s_r[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = logits + (long)b * ld;
  float m = -INFINITY;
  for (int v = tid; v < V; v += 1024) m = fmaxf(m, lg[v]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((tid & 63) == 0) s_r[tid >> 6] = m;
  __syncthreads();
  m = s_r[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, s_r[w]);
  __syncthreads();
  float sum = 0.f;
  for (int v = tid; v < V; v += 1024) sum += expf(lg[v] - m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((tid & 63) == 0) s_r[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += s_r[w];
    p_out[b] = expf(lg[token] - m - logf(tot));
  }
}

}  // namespace

template <bool LN, bool GELU, bool RES, int NW, bool WH = false>
hipError_t sk_launch(dim3 grid, const GemmArgs& g, hipStream_t s) {
  constexpr size_t smem = NW * (SK_WAVE_LDS + 64) * sizeof(float);
  if (smem > 64 * 1024) {        // above the default dynamic-LDS limit (the CU has 160 KB); first called outside any capture
    // the attribute belongs to the CURRENT device's copy of the kernel: once per device, not once per process
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_f32_kernel<LN, GELU, RES, NW, WH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return e;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<LN, GELU, RES, NW, WH>), grid, dim3(64 * NW), smem, s, g);
  return hipGetLastError();
}
template <int NW>
hipError_t sk_dispatch(int kind, dim3 grid, const GemmArgs& g, hipStream_t s) {
  switch (kind) {
    case 0: return sk_launch<false, false, false, NW>(grid, g, s);
    case 1: return sk_launch<false, false, true, NW>(grid, g, s);
    case 2: return sk_launch<false, true, false, NW>(grid, g, s);
    case 3: return sk_launch<false, true, true, NW>(grid, g, s);
    case 4: return sk_launch<true, false, false, NW>(grid, g, s);
    case 5: return sk_launch<true, false, true, NW>(grid, g, s);
    case 6: return sk_launch<true, true, false, NW>(grid, g, s);
    default: return sk_launch<true, true, true, NW>(grid, g, s);
  }
}
template <int TT, bool LN, bool GELU, bool RES, int NW, bool WH>
hipError_t skq_launch(dim3 grid, const GemmArgs& g, hipStream_t s) {
  constexpr size_t smem = NW * (SK_WAVE_LDS + 64) * sizeof(float);
  if (smem > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_q_kernel<TT, LN, GELU, RES, NW, WH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return e;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((gemm_skinny_q_kernel<TT, LN, GELU, RES, NW, WH>), grid, dim3(64 * NW), smem, s, g);
  return hipGetLastError();
}
// the three forms a decode step uses: LayerNorm-folded (q | k | v, cross q), LayerNorm-folded + GELU (MLP first product),
// plain f16 + residual (attention outputs, MLP second product)
template <int TT, int NW>
hipError_t skq_kind(int kind, bool wh, dim3 grid, const GemmArgs& g, hipStream_t s) {
  if (wh) {      // f16 operands: the residual projections; behind an explicit LayerNorm (f16 LayerNorm outputs) bias only and bias + GELU
    if (kind == 1) return skq_launch<TT, false, false, true, NW, true>(grid, g, s);
    if (kind == 0) return skq_launch<TT, false, false, false, NW, true>(grid, g, s);
    if (kind == 2) return skq_launch<TT, false, true, false, NW, true>(grid, g, s);
    return hipErrorInvalidValue;
  }
  if (kind == 4) return skq_launch<TT, true, false, false, NW, false>(grid, g, s);
  if (kind == 6) return skq_launch<TT, true, true, false, NW, false>(grid, g, s);
  return hipErrorInvalidValue;
}
template <int TT>
hipError_t skq_nw(int nw, int kind, bool wh, dim3 grid, const GemmArgs& g, hipStream_t s) {
  switch (nw) {
    case 16: return skq_kind<TT, 16>(kind, wh, grid, g, s);
    case 12: return skq_kind<TT, 12>(kind, wh, grid, g, s);
    case 8: return skq_kind<TT, 8>(kind, wh, grid, g, s);
    default: return skq_kind<TT, 4>(kind, wh, grid, g, s);
  }
}
// true when the resident-weight skinny kernel has a form for this product (the caller de-quantises into its scratch
// slot and takes the dense kernel otherwise)
bool skinny_q_supported(const GemmArgs& g, int batch) {
  if (!(batch == 1 && g.M <= SKINNY_MAX_M && g.K % 128 == 0 && !g.rowtab && !g.tiled)) return false;
  const int kind = (g.ln_s ? 4 : 0) | (g.gelu ? 2 : 0) | (g.residual ? 1 : 0);
  return g.w_half ? (kind == 1 || kind == 0 || kind == 2) : (kind == 4 || kind == 6);
}
hipError_t gemm_skinny_q(const GemmArgs& g, hipStream_t s) {
  const dim3 grid((g.N + 31) / 32, (g.M + 31) / 32);
  const int kind = (g.ln_s ? 4 : 0) | (g.gelu ? 2 : 0) | (g.residual ? 1 : 0);
  const int nw = g.K % 512 == 0 ? 16 : g.K % 384 == 0 ? 12 : g.K % 256 == 0 ? 8 : 4;      // by K alone (see gemm_f32_nt)
  switch (g.wq_type) {
    case QT_Q4_0: return skq_nw<QT_Q4_0>(nw, kind, g.w_half != 0, grid, g, s);
    case QT_Q4_1: return skq_nw<QT_Q4_1>(nw, kind, g.w_half != 0, grid, g, s);
    case QT_Q5_0: return skq_nw<QT_Q5_0>(nw, kind, g.w_half != 0, grid, g, s);
    case QT_Q5_1: return skq_nw<QT_Q5_1>(nw, kind, g.w_half != 0, grid, g, s);
    case QT_Q8_0: return skq_nw<QT_Q8_0>(nw, kind, g.w_half != 0, grid, g, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t gemm_f32_nt(const GemmArgs& g, int batch, hipStream_t s) {
  if (batch == 1 && g.M <= SKINNY_MAX_M && g.K % 128 == 0 && !g.rowtab && !g.tiled) {   // one decode step: latency-bound shape
    const dim3 grid((g.N + 31) / 32, (g.M + 31) / 32);
    const int kind = (g.ln_s ? 4 : 0) | (g.gelu ? 2 : 0) | (g.residual ? 1 : 0);
    // Ways to split K: the widest split the width allows (a wave is left with 1 to 5 chunks: 16 to 80 f32 MFMAs instead
    // of 48 to 160) -- for EVERY row count.  A row's result depends on how K is split (the order its partial sums are
    // added in), so a split chosen by the number of rows made the arithmetic of a clip depend on the size of the batch it
    // was decoded in.  (Rounds 1 - 2 used 4 waves above 64 rows, "the chip is full": measured again in round 3 the wide
    // split is 2 % faster at 128 tiny clips, equal at 256 base clips, 2 % slower at 512 tiny clips.)  With it a clip
    // decodes to the same bits alone, in a batch of 512, and as a row of a multi-position prompt step.
    const int nw = g.K % 512 == 0 ? 16 : g.K % 384 == 0 ? 12 : g.K % 256 == 0 ? 8 : 4;
    if (g.w_half) {                          // f16 weight copy: the residual projections (precision mode 1); bias-only and
                                             // bias + GELU behind an explicit LayerNorm (precision mode 2)
#define CRISPY_SK_WH(GL, RS)                                             \
      switch (nw) {                                                      \
        case 16: return sk_launch<false, GL, RS, 16, true>(grid, g, s);  \
        case 12: return sk_launch<false, GL, RS, 12, true>(grid, g, s);  \
        case 8: return sk_launch<false, GL, RS, 8, true>(grid, g, s);    \
        default: return sk_launch<false, GL, RS, 4, true>(grid, g, s);   \
      }
      if (kind == 1) { CRISPY_SK_WH(false, true) }
      if (kind == 0) { CRISPY_SK_WH(false, false) }
      if (kind == 2) { CRISPY_SK_WH(true, false) }
#undef CRISPY_SK_WH
      return hipErrorInvalidValue;
    }
    switch (nw) {
      case 16: return sk_dispatch<16>(kind, grid, g, s);
      case 12: return sk_dispatch<12>(kind, grid, g, s);
      case 8: return sk_dispatch<8>(kind, grid, g, s);
      default: return sk_dispatch<4>(kind, grid, g, s);
    }
  }
  if (g.w_half) return hipErrorInvalidValue;          // an f16 weight copy is only understood by the skinny kernel
  dim3 grid((g.N + GB_N - 1) / GB_N, (g.M + GB_M - 1) / GB_M, batch);
  hipLaunchKernelGGL(gemm_f32_nt_kernel, grid, dim3(256), 0, s, g);
  return hipGetLastError();
}
hipError_t layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, long rows, int D,
                         hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  switch (D) {     // the widths of the Whisper family (tiny ... large)
    case 384: hipLaunchKernelGGL(layernorm_kernel<6>, grid, block, 0, s, x, gamma, beta, y, rows); break;
    case 512: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, gamma, beta, y, rows); break;
    case 768: hipLaunchKernelGGL(layernorm_kernel<12>, grid, block, 0, s, x, gamma, beta, y, rows); break;
    case 1024: hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, s, x, gamma, beta, y, rows); break;
    case 1280: hipLaunchKernelGGL(layernorm_kernel<20>, grid, block, 0, s, x, gamma, beta, y, rows); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t attn_encoder_f32(const float* qkv, float* out, int B, int T, int D, int heads, hipStream_t s) {
  hipLaunchKernelGGL(attn_enc_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, s, qkv, out, T, D);
  return hipGetLastError();
}
hipError_t attn_decoder_f32(const float* q, long ldq, const float* kv, long kv_batch_stride, long ldkv, long head_stride,
                            long koff, long voff, int n_keys_base, const int* pos_dev, float* out, long ldo, int B, int heads,
                            hipStream_t s, AttnRows rows) {
  if (rows.group < 1) return hipErrorInvalidValue;
  if (rows.stream_kv)
    hipLaunchKernelGGL((attn_dec_kernel<float, true>), dim3(heads, B), dim3(64 * AD_WAVES), 0, s, q, ldq, kv, kv_batch_stride, ldkv, head_stride, koff,
                       voff, n_keys_base, pos_dev, out, ldo, rows);
  else
    hipLaunchKernelGGL((attn_dec_kernel<float, false>), dim3(heads, B), dim3(64 * AD_WAVES), 0, s, q, ldq, kv, kv_batch_stride, ldkv, head_stride, koff,
                       voff, n_keys_base, pos_dev, out, ldo, rows);
  return hipGetLastError();
}
hipError_t attn_decoder_kv16(const float* q, long ldq, const void* kv, long kv_batch_stride, long ldkv, long head_stride,
                             long koff, long voff, int n_keys_base, const int* pos_dev, float* out, long ldo, int B, int heads,
                             hipStream_t s, int max_keys, AttnRows rows) {
  if (rows.group < 1) return hipErrorInvalidValue;
  const int bound = max_keys > 0 ? max_keys : n_keys_base + rows.key_step * (rows.group - 1);
  if ((max_keys > 0 || !pos_dev) && bound <= AD_WAVES * 8 * 12 && AD_WAVES == 16) {   // every byte of K|V requested up front
    const _Float16* kvh = reinterpret_cast<const _Float16*>(kv);
#define CRISPY_ADX(SL) hipLaunchKernelGGL((attn_dec_x16_kernel<SL, false>), dim3(heads, B), dim3(64 * AD_WAVES), 0, s, q, ldq, kvh, \
                                          kv_batch_stride, ldkv, head_stride, koff, voff, n_keys_base, pos_dev, out, ldo, rows)
    // several rows per clip over the same keys (a prompt step's positions, a pass's decoders): the clip's K | V once per chunk of rows
    // (worth it once the one-workgroup-per-row grid is several rounds of the CUs: a row's pass is a latency chain of its own)
    if (rows.group > 1 && rows.key_step == 0 && !rows.key_off && !pos_dev && B % rows.group == 0 && bound > AD_WAVES * 32 && n_keys_base > 0 &&
        (long)heads * B > 512) {
      const dim3 grid(heads, B / rows.group, (rows.group + ADG_ROWS - 1) / ADG_ROWS);
      if (rows.stream_kv)
        hipLaunchKernelGGL((attn_dec_x16g_kernel<true>), grid, dim3(64 * AD_WAVES), 0, s, q, ldq, kvh, kv_batch_stride, ldkv, head_stride, koff,
                           voff, n_keys_base, out, ldo, rows);
      else
        hipLaunchKernelGGL((attn_dec_x16g_kernel<false>), grid, dim3(64 * AD_WAVES), 0, s, q, ldq, kvh, kv_batch_stride, ldkv, head_stride, koff,
                           voff, n_keys_base, out, ldo, rows);
      return hipGetLastError();
    }
    if (rows.stream_kv && bound > AD_WAVES * 32) {     // the cross-attention of a step over many clips
      hipLaunchKernelGGL((attn_dec_x16_kernel<12, true>), dim3(heads, B), dim3(64 * AD_WAVES), 0, s, q, ldq, kvh, kv_batch_stride,
                         ldkv, head_stride, koff, voff, n_keys_base, pos_dev, out, ldo, rows);
      return hipGetLastError();
    }
    if (bound <= AD_WAVES * 8) CRISPY_ADX(1);
    else if (bound <= AD_WAVES * 16) CRISPY_ADX(2);
    else if (bound <= AD_WAVES * 32) CRISPY_ADX(4);
    else CRISPY_ADX(12);
#undef CRISPY_ADX
    return hipGetLastError();
  }
  if (rows.attn16) return hipErrorInvalidValue;      // only the all-keys-up-front kernel above rounds inside the attention
  hipLaunchKernelGGL((attn_dec_kernel<_Float16, false>), dim3(heads, B), dim3(64 * AD_WAVES), 0, s, q, ldq,
                     reinterpret_cast<const _Float16*>(kv), kv_batch_stride, ldkv, head_stride, koff, voff, n_keys_base, pos_dev, out, ldo, rows);
  return hipGetLastError();
}
hipError_t embed_tokens_f32(const int* tokens, const float* tok_emb, const float* pos_emb, int pos, const int* pos_dev,
                            float* x, int B, int D, hipStream_t s, int rows_per_clip, const int* row_off) {
  hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, s, tokens, tok_emb, pos_emb, pos, pos_dev, x, D, rows_per_clip < 1 ? 1 : rows_per_clip, row_off);
  return hipGetLastError();
}
hipError_t argmax_f32(const float* logits, const unsigned char* mask, const unsigned char* mask_first,
                      const int* step_dev, int V, long ld, int* tokens_out, int* tokens_all, float* best, int B, hipStream_t s,
                      int eot, int* finished, int* done_count, const StepFuse* fuse) {
  hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, s, logits, mask, mask_first, step_dev, V, ld, tokens_out,
                     tokens_all, best, eot, finished, done_count, fuse ? *fuse : StepFuse{});
  return hipGetLastError();
}
hipError_t softmax_prob_f32(const float* logits, int V, long ld, int token, float* p_out, int B, hipStream_t s) {
  if (token < 0 || token >= V) return hipErrorInvalidValue;
  hipLaunchKernelGGL(softmax_prob_kernel, dim3(B), dim3(1024), 0, s, logits, V, ld, token, p_out);
  return hipGetLastError();
}

hipError_t beam_kv_reorder(void* kv, void* scratch, const int* parent_dev, int layers, int rows, long row_bytes, long pos_bytes,
                           const int* counters, hipStream_t s) {
  if (pos_bytes % 16 != 0 || row_bytes % 16 != 0) return hipErrorInvalidValue;
  for (int phase = 0; phase < 2; ++phase)
    hipLaunchKernelGGL(beam_kv_copy_kernel, dim3(rows, layers, 8), dim3(256), 0, s, reinterpret_cast<char*>(kv),
                       reinterpret_cast<char*>(scratch), parent_dev, rows, row_bytes, pos_bytes, counters, phase);
  return hipGetLastError();
}

hipError_t beam_advance(const BeamArgs& a, int n_clips, hipStream_t s) {
  if (a.n_dec < 1 || a.n_dec > TS_MAX_CAND || a.n_cand < 1 || a.n_cand > TS_MAX_CAND || a.n_dec * a.n_cand > 64 || n_clips < 1 ||
      a.rows != n_clips * a.n_dec)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_advance_kernel, dim3(n_clips), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t ts_pick(const TsPickArgs& a, int B, hipStream_t s) {
  if (a.n_cand < 0 || a.n_cand > TS_MAX_CAND || (a.n_cand > 0 && !(a.u_all && a.cand_tok && a.cand_plog && a.cand_tid))) return hipErrorInvalidValue;
  if (a.u_all) {
    if (!a.temperature || !a.x_scratch || a.V > TS_BLK * TS_NB) return hipErrorInvalidValue;      // a thread's ids: TS_NB runs of TS_E
    hipLaunchKernelGGL(ts_sample_kernel, dim3(B), dim3(TS_THREADS), 0, s, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(ts_pick_kernel, dim3(B), dim3(1024), 0, s, a);
  return hipGetLastError();
}

}  // namespace crispy
