// whisper_dec_gemv.hip -- the projections of a generated token's decoder step for the CATALOG widths (768 / 1024 / 1280:
// small, medium, large-v3 / turbo -- src-tauri/src/managers/model.rs:74-148), dense f16 or resident ggml blocks, at the
// reference's call shape: one chunk at a time (engine.transcribe, managers/transcription.rs:183-185), i.e. 1 .. 4 rows.
//
// Round 5 ran these steps through the 32 x 32-tile "skinny" kernels (whisper_kernels.hip): N / 32 workgroups -- 32 of the
// chip's 256 CUs for a 1024-wide projection -- each walking K in chunks of 32 per wave, plus a LayerNorm launch in front of
// three of the six projections: 11 launches per layer, 4.5 - 6 us each whatever they do (15 us for fc2, K = 4096 walked by
// 16 waves), 1.75 - 1.94 ms per token on Whisper-medium (profiles/r06_decode_medium_*).  A step of one row is a
// matrix-VECTOR product per projection; what it needs is every weight byte requested at once, from as many CUs as there are:
//
//   * one weight row per HALF-WAVE, a lane owns whole 32-weight blocks of it (= one ggml block of a resident model; 64
//     contiguous bytes of a dense f16 row): every request of the row is issued in the kernel's first instructions;
//   * 8 weight rows per 256-thread workgroup: N / 8 workgroups (128 for a 1024-wide projection, 512 for fc1);
//   * the LayerNorm in front of q | k | v, cross-q and fc1 is computed by every workgroup itself (a row is <= 1280 floats:
//     cheaper than the launch it replaces), the activation rows live in LDS as f16 -- ggml's mul_mat operand;
//   * resident blocks are de-quantised in registers with the loader's operations in the loader's order (asr_quant.h), then
//     rounded to f16: a resident model and the same file inflated at load run the SAME instructions behind the fetch, lane
//     for lane and block for block, so they agree bit for bit (tests/test_gpu_resident.py, test_catalog_models_at_full_depth);
//   * a row's arithmetic involves that row alone: its bits do not depend on how many rows share the step.
//
// Arithmetic = ggml's for these products [UPSTREAM-RECALL: mul_mat converts its f32 operand to the f16 of the weight]: the
// f32 activation (or LayerNorm output) rounded to f16 against f16 weights, f32 accumulation (v_dot2_f32_f16 chains of 16
// pairs per block, blocks of a lane in K order, then a fixed shuffle tree over the 32 lanes of the row).
#include "asr_common.h"
#include "asr_quant.h"

namespace crispy {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int GV_THREADS = 256;
constexpr int GV_ROWS = 8;                      // weight rows per workgroup: two per wave, one per half-wave
constexpr int GV_XB = 40;                       // halves between two 32-element blocks of an activation row in LDS (32 + 8: a
                                                // half-wave's 64-byte reads, one block per lane, spread over all banks)

__device__ __forceinline__ float gv_dot8(const half8 a, const half8 b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2(half2v{a[2 * i], a[2 * i + 1]}, half2v{b[2 * i], b[2 * i + 1]}, acc, false);
  return acc;
}

// ---- one block of 32 weights of one row: requested as raw registers, decoded to 4 x half8 (k order) at the point of use ----
template <int TT> struct GvBlock {                  // ggml block (asr_quant.h)
  static constexpr int BB = quant_block_bytes(TT);
  static constexpr int NWD = (BB + 3) / 4;          // (the last dword may reach 2 bytes past the block: tensors are padded)
  unsigned w[NWD];
  __device__ __forceinline__ void request(const unsigned char* row, int kb) {
    const unsigned char* b = row + (long)kb * BB;
#pragma unroll
    for (int i = 0; i < NWD; ++i) w[i] = q_u32(b + 4 * i);
  }
  __device__ __forceinline__ unsigned byte_at(int off) const {      // byte `off` of the block (compile-time off)
    return (w[off >> 2] >> (8 * (off & 3))) & 0xffu;
  }
  __device__ __forceinline__ float half_at(int off) const {         // f16 at byte `off` (even)
    const unsigned short u = (unsigned short)((w[off >> 2] >> (8 * (off & 3))) & 0xffffu);
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
  }
  __device__ __forceinline__ unsigned dword_at(int off) const {     // 32 bits from byte `off` (2-byte aligned)
    if ((off & 3) == 0) return w[off >> 2];
    return (w[off >> 2] >> 16) | (w[(off >> 2) + 1] << 16);
  }
  // The 32 weights as f16, computed with the loader's operations in the loader's order (int -> float, one multiply, one add,
  // each rounded on its own: asr_quant.h q_block), arranged for few instructions: the sixteen quant bytes as four aligned
  // dwords, low and high nibbles masked four at a time, a q5 block's fifth bits spread onto them by one multiply per four
  // elements, bytes converted by v_cvt_f32_ubyteN; (float)(x - 8) is written (float)x - 8.0f (both exact).
  __device__ __forceinline__ void decode(half8 (&h)[4]) const {
#pragma clang fp contract(off)
    float y[32];
    const float d = half_at(0);
    if (TT == QT_Q8_0) {
#pragma unroll
      for (int j = 0; j < 32; ++j) y[j] = (float)(int)(signed char)byte_at(2 + j) * d;
    } else {
      constexpr bool has_m = TT == QT_Q4_1 || TT == QT_Q5_1, has_h = TT == QT_Q5_0 || TT == QT_Q5_1;
      const float m = has_m ? half_at(2) : 0.f;
      constexpr int off_h = has_m ? 4 : 2;
      const unsigned qh = has_h ? dword_at(off_h) : 0u;
      constexpr int off_q = off_h + (has_h ? 4 : 0);
      constexpr float zero = TT == QT_Q4_0 ? 8.f : 16.f;            // q4_0 / q5_0: the integer offset
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned q = dword_at(off_q + 4 * i);                  // elements 4 i .. 4 i + 3 (low nibbles) and 16 + 4 i .. (high)
        unsigned lo = q & 0x0f0f0f0fu, hi = (q >> 4) & 0x0f0f0f0fu;
        if (has_h) {
          lo |= ((((qh >> (4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4;
          hi |= ((((qh >> (16 + 4 * i)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = (float)((lo >> (8 * e)) & 0xffu), x1 = (float)((hi >> (8 * e)) & 0xffu);
          if (has_m) { y[4 * i + e] = x0 * d + m; y[16 + 4 * i + e] = x1 * d + m; }
          else { y[4 * i + e] = (x0 - zero) * d; y[16 + 4 * i + e] = (x1 - zero) * d; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)y[8 * i + e];
  }
};
template <> struct GvBlock<-1> {                    // dense f16 row: 64 bytes
  static constexpr int BB = 64;
  half8 h4[4];
  __device__ __forceinline__ void request(const unsigned char* row, int kb) {
    const half8* p = reinterpret_cast<const half8*>(row + (long)kb * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) h4[i] = p[i];
  }
  __device__ __forceinline__ void decode(half8 (&h)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = h4[i];
  }
};

// K -> LDS index of an activation row held as f16 in blocks of 32 with 8 halves of padding
__device__ __forceinline__ int gv_idx(int c) { return (c >> 5) * GV_XB + (c & 31); }

// One wave, one row: f16(LayerNorm(x)) into the padded LDS row.  Two passes over registers, a lane holds columns lane + 64 q
// (layernorm_h_kernel's arithmetic: the sum, then the sum of squared deviations, shuffle trees of the same shape).
// EARLY_GB: gamma and beta requested with the row (3 D / 64 registers: the 256-thread kernels have them to spare); else read
// when they are multiplied (the 1024-thread cross kernel lives in 128 registers).
template <int D, bool EARLY_GB = true>
__device__ __forceinline__ void gv_layernorm_wave(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  _Float16* out, int lane) {
  constexpr int PER = D / 64;
  float e[PER], gm[EARLY_GB ? PER : 1], bt[EARLY_GB ? PER : 1], s = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    e[q] = x[lane + 64 * q];
    if (EARLY_GB) { gm[q] = gamma[lane + 64 * q]; bt[q] = beta[lane + 64 * q]; }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) s += e[q];
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
  for (int q = 0; q < PER; ++q) {
    const float g = EARLY_GB ? gm[q] : gamma[lane + 64 * q], b = EARLY_GB ? bt[q] : beta[lane + 64 * q];
    out[gv_idx(lane + 64 * q)] = (_Float16)((e[q] - mean) * rstd * g + b);
  }
}

// TT: -1 dense f16 rows, else the ggml type.  K: length of a weight row.  LN: the activation is LayerNorm(x) (K = model width).
// EPI: GEMV_QKV / GEMV_RES / GEMV_F32 / GEMV_GELU16 (asr_common.h).
template <int TT, int K, bool LN, int EPI>
__global__ __launch_bounds__(GV_THREADS) void gemv_dec_kernel(GemvArgs g) {
  constexpr int KB = K / 32;                      // blocks per weight row
  constexpr int NPASS = (KB + 31) / 32;           // blocks per lane (lane hl of the half-wave owns blocks hl, hl + 32, ...)
  __shared__ __attribute__((aligned(16))) _Float16 xs[GEMV_MAX_M][KB * GV_XB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane & 31, half = lane >> 5;
  const int n = blockIdx.x * GV_ROWS + 2 * wave + half;              // this half-wave's weight row (N is a multiple of 8)
  // Steps of more than GEMV_MAX_M rows: the workgroup takes the rows four at a time over the SAME decoded weights (blockIdx.y
  // deals the chunks when the weight-row blocks alone do not fill the chip).  A row's arithmetic is what it is in a step of
  // its own -- one clip, one answer at every batch size: the reason the wide steps of the catalog widths run here too and
  // not on the MFMA tiles of the skinny kernels.
  // (1) every weight byte of the row, requested first
  const unsigned char* wrow;
  if (TT < 0) {
    wrow = reinterpret_cast<const unsigned char*>(g.w16) + (long)n * K * 2;
  } else {
    const int p = n / g.wq_rows;                                      // part of a row-fused matrix (q | k | v: three tensors)
    const unsigned char* base = p == 0 ? g.wq[0] : (p == 1 ? g.wq[1] : g.wq[2]);
    wrow = base + (long)(n - p * g.wq_rows) * KB * GvBlock<TT>::BB;
  }
  GvBlock<TT> blk[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int kb = min(hl + 32 * ps, KB - 1);                         // lanes past the row's end: its last block again, weight 0 below
    blk[ps].request(wrow, kb);
  }
  const float bias = g.bias ? g.bias[n] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  half8 w[NPASS][4];
  bool decoded = false;
  // plain activation rows (f16 behind a LayerNorm launch, the GELU'd hidden units, f32 attention outputs): the NEXT chunk's
  // values are requested before this chunk's products, so that a wide step does not wait a global round trip per four rows
  constexpr bool PLAIN = !LN && EPI != GEMV_RES_MERGE;
  constexpr int XPT = (K + GV_THREADS - 1) / GV_THREADS;          // columns per thread and row: c = tid + GV_THREADS q
  _Float16 nx[PLAIN ? GEMV_MAX_M : 1][PLAIN ? XPT : 1];
  auto fetch = [&](int r0) {
    if constexpr (PLAIN) {
      const int M = min(GEMV_MAX_M, g.M - r0);
#pragma unroll
      for (int m = 0; m < GEMV_MAX_M; ++m) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
          const int c = min(tid + GV_THREADS * q, K - 1);
          const long at = (long)(r0 + min(m, M - 1)) * g.ldx + c;
          nx[m][q] = g.x16 ? g.x16[at] : (_Float16)g.x[at];
        }
      }
    }
  };
  const int r_step = GEMV_MAX_M * (int)gridDim.y;
  if (blockIdx.y * GEMV_MAX_M < g.M) fetch(blockIdx.y * GEMV_MAX_M);
  for (int r0 = blockIdx.y * GEMV_MAX_M; r0 < g.M; r0 += r_step) {
    const int M = min(GEMV_MAX_M, g.M - r0);
    // (2) the chunk's activation rows as f16 in LDS
    if (LN) {
      if (wave < M) gv_layernorm_wave<K>(g.x + (long)(r0 + wave) * g.ldx, g.ln_g, g.ln_b, xs[wave], lane);
    } else if (EPI == GEMV_RES_MERGE) {
      // the activation is the cross-attention output, still in XA_PARTS partial soft-maxes per head (gv_xattn_kernel): merged
      // here, in a fixed order, by every workgroup for itself (17 KB of partials per row against the launch it saves)
      for (int m = 0; m < M; ++m) {
        for (int c = tid; c < K; c += GV_THREADS) {
          const float* ph = g.xpart + ((long)(r0 + m) * (K / 64) + (c >> 6)) * (XA_PARTS * XA_PART_FLOATS);
          float mx = ph[0];
#pragma unroll
          for (int j = 1; j < XA_PARTS; ++j) mx = fmaxf(mx, ph[j * XA_PART_FLOATS]);
          float num = 0.f, den = 0.f;
#pragma unroll
          for (int j = 0; j < XA_PARTS; ++j) {
            const float scl = __expf(ph[j * XA_PART_FLOATS] - mx);         // an empty part has m = -1e30: scale 0
            num = fmaf(ph[j * XA_PART_FLOATS + 2 + (c & 63)], scl, num);
            den = fmaf(ph[j * XA_PART_FLOATS + 1], scl, den);
          }
          xs[m][gv_idx(c)] = (_Float16)(num / den);
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < GEMV_MAX_M; ++m) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
          const int c = tid + GV_THREADS * q;
          if (m < M && c < K) xs[m][gv_idx(c)] = nx[m][q];
        }
      }
      if (r0 + r_step < g.M) fetch(r0 + r_step);
    }
    __syncthreads();
    if (!decoded) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) blk[ps].decode(w[ps]);
      decoded = true;
    }
    // (3) the products: a lane's blocks in K order, 16 dot2 per block and row
    float acc[GEMV_MAX_M];
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) acc[m] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int kb = hl + 32 * ps;
      if (kb < KB) {
#pragma unroll
        for (int m = 0; m < GEMV_MAX_M; ++m) {
          if (m < M) {
            const half8* xp = reinterpret_cast<const half8*>(&xs[m][kb * GV_XB]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m] = gv_dot8(w[ps][i], xp[i], acc[m]);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) acc[m] += __shfl_xor(acc[m], off, 64);      // within the half-wave: fixed tree
    }
    // (4) epilogue: lane 0 of the half-wave owns output column n of every row
    if (hl == 0) {
#pragma unroll
      for (int m = 0; m < GEMV_MAX_M; ++m) {
        if (m >= M) break;
        const long row = r0 + m;
        float v = acc[m] + bias;
        if (EPI == GEMV_QKV) {
          const int D = K;
          if (n < D) {
            g.out[row * g.ldo + n] = v;
          } else {                                                       // k | v of this position into the f16 cache row
            const long pos = g.pos_dev ? (long)*g.pos_dev : (long)g.pos;
            g.kv[row * g.kv_row_stride + pos * (2L * D) + (n - D)] = (_Float16)v;
          }
        } else if (EPI == GEMV_RES || EPI == GEMV_RES_MERGE) {
          g.out[row * g.ldo + n] = v + g.res[row * g.ldo + n];
        } else if (EPI == GEMV_F32) {
          g.out[row * g.ldo + n] = v;
        } else {
          g.out16[row * g.ldo + n] = (_Float16)gelu_ggml(v);
        }
      }
    }
    __syncthreads();                                                   // xs is rewritten by the next chunk
  }
}

// ---- cross-attention of a decode step: q of one (row, head) against a QUARTER of the clip's keys ----
// Round 5's cross block here -- one workgroup per head streaming the head's 384 KB of K | V (10.5 us at width 1024: sixteen
// CUs pull everything) -- becomes H x XA_PARTS workgroups per row: every workgroup attends its quarter of the keys and leaves an
// UNNORMALISED partial soft-max (maximum, sum, 64 weighted value sums) that the output projection merges in its prologue
// (GEMV_RES_MERGE).  Partition inside a workgroup: 16 waves x 3 slots x 8 keys (8 lanes per key row of 64 halves), the
// arithmetic of attn_dec_x16_kernel per wave; merges in fixed order.  (First built with 4 waves x 12 slots: 11 us per launch
// -- a wave's chain of 12 score / value slots is what the launch lasts, so the chains were cut by four.)
// q comes from a gemv_dec launch of its own (GEMV_F32 over LayerNorm(x)).  The first form projected it here, each workgroup its
// head's 64 rows behind its own LayerNorm, to save that launch -- and read the head's weights (128 KB as f16) once per quarter:
// 13.5 - 16 us per launch at ONE row against 5 + ~6 for the pair (medium-q4_1: 1.13 -> 1.01 ms per position), 178 us at 64 rows.
constexpr int XA_THREADS = 1024;                          // 16 waves: a part's 375 keys are 3 slots of 8 per wave
template <int D>
__global__ __launch_bounds__(XA_THREADS) void gv_xattn_kernel(XattnArgs a) {
  constexpr int XW = XA_THREADS / 64;                     // waves
  static_assert(XW * XA_SLOTS * 8 >= 384, "partition");
  __shared__ __attribute__((aligned(16))) float q_s[64];
  __shared__ __attribute__((aligned(16))) float part_o[XW][64];
  __shared__ float part_m[XW], part_l[XW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x / XA_PARTS, part = blockIdx.x % XA_PARTS, row = blockIdx.y;
  const int clip = row / a.group;
  // (1) this workgroup's keys and values, requested first: wave w takes keys [k_lo, k_hi) of the part
  const int Tn = a.n_keys;
  const int per_p = (Tn + XA_PARTS - 1) / XA_PARTS;
  const int p_lo = part * per_p, p_hi = min(Tn, p_lo + per_p);
  const int per = (max(p_hi - p_lo, 0) + XW - 1) / XW;
  const int k_lo = p_lo + wave * per, k_hi = min(p_hi, k_lo + per);
  const int k_last = min(max(k_hi - 1, 0), Tn - 1);
  const int c8 = lane & 7, r8 = lane >> 3;
  const char* Kb = reinterpret_cast<const char*>(a.xkv + (long)clip * a.clip_stride + (long)h * 64 * Tn);
  const char* Vb = Kb + (long)Tn * D * 2;
  half8 kr[XA_SLOTS], vr[XA_SLOTS];
#pragma unroll
  for (int i = 0; i < XA_SLOTS; ++i) {
    const unsigned off = (unsigned)((min(k_lo + 8 * i + r8, k_last) * 64 + 8 * c8) * 2);
    kr[i] = *reinterpret_cast<const half8*>(Kb + off);
    vr[i] = *reinterpret_cast<const half8*>(Vb + off);
  }
  // (2) q of the head
  if (tid < 64) q_s[tid] = a.q[(long)row * a.ldq + h * 64 + tid];
  __syncthreads();
  // (3) scores, soft-max weights and weighted value sums of this wave's keys (attn_dec_x16_kernel's arithmetic)
  float qv[8];
  {
    const float4 q0 = *reinterpret_cast<const float4*>(q_s + 8 * c8);
    const float4 q1 = *reinterpret_cast<const float4*>(q_s + 8 * c8 + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w;
    qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
  }
  float sc[XA_SLOTS], mloc = -1e30f;
#pragma unroll
  for (int i = 0; i < XA_SLOTS; ++i) {
    float v = (float)kr[i][0] * qv[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) v = fmaf((float)kr[i][e], qv[e], v);
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    sc[i] = k_lo + 8 * i + r8 < k_hi ? v : -1e30f;
    mloc = fmaxf(mloc, sc[i]);
  }
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
  float lsum = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < XA_SLOTS; ++i) {
    const float pw = k_lo + 8 * i + r8 < k_hi ? __expf(sc[i] - mloc) : 0.f;
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
  if (r8 == 0) {
    *reinterpret_cast<float4*>(&part_o[wave][8 * c8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(&part_o[wave][8 * c8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  if (lane == 0) { part_m[wave] = mloc; part_l[wave] = lsum; }
  __syncthreads();
  // (4) the four waves' partials into one (still unnormalised), for the output projection to merge with the other parts
  if (wave == 0) {
    float m = part_m[0];
#pragma unroll
    for (int w = 1; w < XW; ++w) m = fmaxf(m, part_m[w]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < XW; ++w) {
      const float scl = __expf(part_m[w] - m);
      o = fmaf(part_o[w][lane], scl, o);
      l = fmaf(part_l[w], scl, l);
    }
    float* dst = a.part + (((long)row * (D / 64) + h) * XA_PARTS + part) * XA_PART_FLOATS;
    dst[2 + lane] = o;
    if (lane == 0) { dst[0] = m; dst[1] = l; }
  }
}

template <int TT, int K, bool LN, int EPI>
hipError_t gv_launch(const GemvArgs& g, hipStream_t s) {
  // the rows' chunks of four over gridDim.y only as far as the weight-row blocks alone leave the chip empty (~ 1024 workgroups)
  const int chunks = (g.M + GEMV_MAX_M - 1) / GEMV_MAX_M, xb = g.N / GV_ROWS;
  const int gy = std::max(1, std::min(chunks, (1024 + xb - 1) / xb));
  hipLaunchKernelGGL((gemv_dec_kernel<TT, K, LN, EPI>), dim3(xb, gy), dim3(GV_THREADS), 0, s, g);
  return hipGetLastError();
}
template <int K, bool LN, int EPI>
hipError_t gv_by_type(const GemvArgs& g, hipStream_t s) {
  if (g.w16) return gv_launch<-1, K, LN, EPI>(g, s);
  switch (g.wq_type) {
    case QT_Q4_0: return gv_launch<QT_Q4_0, K, LN, EPI>(g, s);
    case QT_Q4_1: return gv_launch<QT_Q4_1, K, LN, EPI>(g, s);
    case QT_Q5_0: return gv_launch<QT_Q5_0, K, LN, EPI>(g, s);
    case QT_Q5_1: return gv_launch<QT_Q5_1, K, LN, EPI>(g, s);
    case QT_Q8_0: return gv_launch<QT_Q8_0, K, LN, EPI>(g, s);
    default: return hipErrorInvalidValue;
  }
}
template <bool LN, int EPI>
hipError_t gv_by_width(const GemvArgs& g, hipStream_t s) {
  switch (g.K) {
    case 768: return gv_by_type<768, LN, EPI>(g, s);
    case 1024: return gv_by_type<1024, LN, EPI>(g, s);
    case 1280: return gv_by_type<1280, LN, EPI>(g, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool gemv_dec_supported(int D, int rows) { return (D == 768 || D == 1024 || D == 1280) && rows >= 1 && rows <= GEMV_MAX_ROWS; }

namespace {
}  // namespace

hipError_t gemv_xattn(const XattnArgs& a, hipStream_t s) {
  if (a.rows < 1 || a.rows > GEMV_MAX_ROWS || a.group < 1 || a.n_keys < 1 || a.n_keys > XA_PARTS * (XA_THREADS / 64) * XA_SLOTS * 8 || !a.q)
    return hipErrorInvalidValue;
  const dim3 grid((unsigned)(a.D / 64 * XA_PARTS), (unsigned)a.rows), block(XA_THREADS);
  switch (a.D) {
    case 768: hipLaunchKernelGGL((gv_xattn_kernel<768>), grid, block, 0, s, a); break;
    case 1024: hipLaunchKernelGGL((gv_xattn_kernel<1024>), grid, block, 0, s, a); break;
    case 1280: hipLaunchKernelGGL((gv_xattn_kernel<1280>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t gemv_dec(const GemvArgs& g, int epi, hipStream_t s) {
  if (g.M < 1 || g.M > GEMV_MAX_ROWS || g.N % GV_ROWS != 0 || (!g.w16 && !g.wq[0])) return hipErrorInvalidValue;
  if (!g.w16 && (g.wq_rows <= 0 || g.wq_rows % GV_ROWS != 0)) return hipErrorInvalidValue;
  const bool ln = g.ln_g != nullptr;
  switch (epi) {
    case GEMV_QKV: return ln ? gv_by_width<true, GEMV_QKV>(g, s) : (g.x16 ? gv_by_width<false, GEMV_QKV>(g, s) : hipErrorInvalidValue);
    case GEMV_F32: return ln ? gv_by_width<true, GEMV_F32>(g, s) : (g.x16 ? gv_by_width<false, GEMV_F32>(g, s) : hipErrorInvalidValue);
    case GEMV_GELU16: return ln ? gv_by_width<true, GEMV_GELU16>(g, s) : (g.x16 ? gv_by_width<false, GEMV_GELU16>(g, s) : hipErrorInvalidValue);
    case GEMV_RES_MERGE:
      if (ln || !g.xpart) return hipErrorInvalidValue;
      switch (g.K) {
        case 768: return gv_by_type<768, false, GEMV_RES_MERGE>(g, s);
        case 1024: return gv_by_type<1024, false, GEMV_RES_MERGE>(g, s);
        case 1280: return gv_by_type<1280, false, GEMV_RES_MERGE>(g, s);
        default: return hipErrorInvalidValue;
      }
    case GEMV_RES:
      if (ln) return hipErrorInvalidValue;
      switch (g.K) {                    // out-projections (K = D) and fc2 (K = 4 D)
        case 768: return gv_by_type<768, false, GEMV_RES>(g, s);
        case 1024: return gv_by_type<1024, false, GEMV_RES>(g, s);
        case 1280: return gv_by_type<1280, false, GEMV_RES>(g, s);
        case 3072: return gv_by_type<3072, false, GEMV_RES>(g, s);
        case 4096: return gv_by_type<4096, false, GEMV_RES>(g, s);
        case 5120: return gv_by_type<5120, false, GEMV_RES>(g, s);
        default: return hipErrorInvalidValue;
      }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace crispy
