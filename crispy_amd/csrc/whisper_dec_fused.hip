// whisper_dec_fused.hip -- a generated token's decoder layer in THREE launches instead of eight (precision modes 1 / 2,
// dense f16 weight copies, widths 384 / 512: Whisper-tiny / -base, the models of BASELINE.json's configs).
// Reference call shape: engine.transcribe(&audio, &TranscribeOptions::default()), one clip, token by token
// (src-tauri/src/managers/transcription.rs:183-185); whisper.cpp's decoder graph per token [UPSTREAM-RECALL].
//
// Why not one persistent launch with grid barriers: a dependent kernel boundary costs 1.2 - 1.9 us on
// this chip, a device-wide barrier 4 - 7 us and a flagged cross-CU hand-off 1.3 - 5 us (MI355X_MICROARCH.md, price list:
// boundary / barrier-xcd / handoff-flag) -- replacing boundaries by barriers loses.  What a step of ~40 launches of
// 5 - 6 us each pays for is (a) the launches that exist only because a stage's output has to be COMPLETE in memory before
// the next stage may read it, and (b) the weight fetch of every stage starting only when the stage does.  So:
//
//   * a stage's all-to-all (every output column needs every head / every hidden unit) is not waited for: the producer
//     writes one PARTIAL row per head (attention output projections) or per 128 hidden units (MLP), and the consumer adds
//     them up -- in a fixed order, so a row's bits never depend on what else is in the batch -- while it assembles the
//     residual stream it needs anyway.  The boundary between "projection" and "what reads it" disappears:
//
//       fused_self_kernel   x0 = x + b + sum(partials);  LayerNorm;  q | k | v of ONE head;  k, v -> f16 cache;  causal attention over
//                           the cache;  partial out-projection of that head                        grid (heads, rows)
//       fused_cross_kernel  x1 = x0 + b + sum(head partials);  LayerNorm;  q of one head;  attention over the clip's cross
//                           K | V (f16, streamed into registers);  partial out-projection       grid (heads, rows)
//       fused_mlp_kernel    x2 = x1 + b + sum(head partials);  LayerNorm;  128 hidden units: fc1 + ggml GELU;  their partial
//                           fc2                                                                   grid (4 D / 128, rows)
//       fused_finish_kernel x = x2 + b + sum(MLP partials);  final LayerNorm -> f16 row for the vocabulary projection
//                           (steps of <= 4 rows: the vocabulary kernel does this itself, whisper_dec_f16.hip / fd_ln.h)
//
//   * one workgroup = one head (or 128 hidden units) of 1 - 8 rows, 16 waves; every weight byte the workgroup needs
//     (196 KB at D = 384) is requested in its first instructions, together with the residual stream and the self K | V
//     cache, so the chain LayerNorm -> product -> attention -> product inside a launch waits for memory ONCE.  The
//     products of the self and MLP blocks run on v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand, straight from
//     the registers they were requested into (tile = 16 weight rows, lane = row l & 15, k = 8 (l >> 4) .. + 7), and the
//     rows of the step as the 16 columns of B (from LDS): a column's result depends on that column alone, so a row
//     decodes to the same bits whatever else is in the workgroup or the batch, and 1 .. 16 rows cost the same.  (The first
//     form multiplied on v_dot2c_f32_f16 with DPP row sums: with 4 rows per workgroup its 16 waves spent 5 us in the
//     vector pipe.)  The cross block is one row per workgroup by construction (its K | V fills the registers) and keeps
//     the matrix-VECTOR form.
//
// Arithmetic = ggml's for these products [UPSTREAM-RECALL: mul_mat converts its f32 operand to the f16 of the weight]:
// LayerNorm in f32, its output rounded to f16 against f16 weights, f32 accumulation; q . k with the f32 query against the f16
// cache (mode 2: the query and the normalised probabilities rounded to f16, AttnRows::attn16); attention output and GELU'd
// hidden units rounded to f16 against f16 weights.  Oracle: oracle/whisper_oracle.py DecoderCache(f16=True, ln16=True).
#include "asr_common.h"
#include "fd_ln.h"

namespace crispy {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FD_NB = 16;          // columns of an MFMA tile = rows of the step a workgroup can take

constexpr int FD_WAVES = 16;
constexpr int FD_THREADS = 64 * FD_WAVES;

__device__ __forceinline__ float dot8(const half8 a, const half8 b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2(half2v{a[2 * i], a[2 * i + 1]}, half2v{b[2 * i], b[2 * i + 1]}, acc, false);
  return acc;
}
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 8 / 16 lanes that share a row (every one of them ends up with the total)
__device__ __forceinline__ float sum8(float v) { return dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v))); }
__device__ __forceinline__ float sum16(float v) { return dpp_add<0x140>(sum8(v)); }

// Workgroup barrier for LDS traffic ONLY: __syncthreads() waits for every outstanding vector-memory operation first
// (s_waitcnt vmcnt(0)), i.e. the first barrier of a kernel would wait for all the weights and keys requested at its top
// and serialise exactly the overlap these kernels are built on (measured: a row's pass through the MLP block cost ~3 us
// with the next row's residual stream "in flight").  Nothing in these kernels hands GLOBAL data from wave to wave.
__device__ __forceinline__ void fd_bar() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// a load from a workgroup-uniform base (scalar registers) + a 32-bit byte offset per lane: one address register instead of
// two per request -- these kernels keep up to 60 requests per lane in flight
template <class T> __device__ __forceinline__ T ldu(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// The residual stream entering a block, and its LayerNorm as the f16 vector the products read:
//   xs = x_in + bias + part[0] + part[1] + ... (this order), written to x_out by the row's first workgroup;
//   xn = f16(LayerNorm(xs)) -- one wave, a lane holds columns lane + 64 q, two passes (layernorm_h_kernel's arithmetic).
// RB rows per workgroup: thread tid holds column tid % D of rows tid / D, tid / D + S, ... (S = 1024 / D = 2 rows per pass).
template <int D, int NP>
struct FdParams {          // what does not depend on the row: LayerNorm gamma | beta (staged in LDS), the bias
  float gm, bt, b;
  __device__ __forceinline__ void request(const FusedIn& in) {
    const int col = threadIdx.x % D;
    gm = in.ln_g[col];
    bt = in.ln_b[col];
    b = NP > 0 ? in.bias[col] : 0.f;
  }
  __device__ __forceinline__ void stage(float* gb) const {      // visible to the LayerNorm waves behind the first finish()'s barrier
    const int tid = threadIdx.x;
    if (tid < D) { gb[tid] = gm; gb[D + tid] = bt; }
  }
};
template <int D, int NP, int RB>
struct FdInput {
  static constexpr int S = FD_THREADS / D;
  static constexpr int PASSES = (RB + S - 1) / S;
  float v[PASSES], pv[PASSES][NP > 0 ? NP : 1];
  // the loads, issued FIRST in a kernel: they return first, and the LayerNorm runs while the weights are still arriving
  __device__ __forceinline__ void request(const FusedIn& in, int rows, int row0) {
    const int tid = threadIdx.x, slot = tid / D, col = tid % D;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int r = ps * S + slot;
      if (slot < S && r < RB) {
        const int row = min(row0 + r, rows - 1);                 // past the last row: a valid row again, never stored
        v[ps] = ldu<float>(in.x_in + (long)row * D, 4u * col);
        if (NP > 0) {
          const float* prow = in.part + (long)row * D;
#pragma unroll
          for (int p = 0; p < NP; ++p) pv[ps][p] = ldu<float>(prow, 4u * (unsigned)(p * rows * D + col));
        }
      }
    }
  }
  // xs [RB][D] f32, gb [2][D] (FdParams::stage), xn [RB][D] f16.  Row r's LayerNorm is wave r's.
  // XLD: halves between two rows of xn (D for the matrix-vector readers; D + 8 where xn is an MFMA operand: 16 rows read
  // side by side then fall on different banks)
  template <int XLD = D>
  __device__ __forceinline__ void finish(const FusedIn& in, float bias, int rows, int row0, bool writer, float* xs, const float* gb,
                                         _Float16* xn) {
    const int tid = threadIdx.x, slot = tid / D, col = tid % D;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int r = ps * S + slot;
      if (slot < S && r < RB) {
        const float x = fd_assemble<NP>(v[ps], bias, pv[ps]);
        xs[r * D + col] = x;
        if (writer && row0 + r < rows) in.x_out[(long)(row0 + r) * D + col] = x;
      }
    }
    fd_bar();
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < RB) fd_layernorm_wave<D>(xs + wave * D, gb, xn + wave * XLD, lane);
    fd_bar();
  }
};

// Partial out-projection of one head: po[n] = sum_j W[n][col0 + j] f16(att[j]), n < D, j < 64.  8 lanes per row (one
// 128-byte line), 128 rows per pass of the workgroup.  request() early, finish() once att_h is in LDS.
template <int D>
struct HeadOut {
  static constexpr int NPASS = D / 128;
  half8 w[NPASS];
  __device__ __forceinline__ void request(const _Float16* __restrict__ W, int col0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
      w[p] = ldu<half8>(W + col0, 2u * (unsigned)((128 * p + 8 * wave + (lane >> 3)) * D + 8 * (lane & 7)));
  }
  __device__ __forceinline__ void finish(const _Float16* att_h, float* po) const {      // att_h [64], po [D]: one row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const half8 a = *reinterpret_cast<const half8*>(att_h + 8 * (lane & 7));
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const float v = sum8(dot8(w[p], a, 0.f));
      if ((lane & 7) == 0) po[128 * p + 8 * wave + (lane >> 3)] = v;
    }
  }
};

// One query against the f16 keys / values of ONE head held in registers: SLOTS slots of 8 keys per wave (8 lanes per key
// row of 64 halves), the 16 waves split the keys, partial (max, sum, P.V) triples meet in LDS -- the partition, the
// arithmetic and the merge of attn_dec_x16_kernel (whisper_kernels.hip), so a row decodes to the same bits whichever
// launch form ran its attention.  valid key <=> k_lo + 8 i + r < k_hi.  Returns with att_h[0..63] written (f16 of the
// attention output) and a workgroup barrier behind it.
// mid(): called once the keys and values have been consumed (their registers are free) and before the merge: the place
// to request what comes after the attention.
// after_scores(): called when every key has been consumed -- the cross-attention requests its values there, into the
// registers the keys leave (both at once do not fit beside the projections: 48 + 48 of 128 registers).
// MERGE = false: stops once the wave's partial (max, sum, P.V) is in part_o / part_m / part_l -- the caller runs the
// partials of several rows, ONE barrier, and fd_merge of row r on wave r (the self block with RB rows).
__device__ __forceinline__ void fd_merge(const float (*part_o)[64], const float* part_m, const float* part_l, int attn16, _Float16* att_h) {
  const int lane = threadIdx.x & 63;
  float m = part_m[0];
#pragma unroll
  for (int w = 1; w < FD_WAVES; ++w) m = fmaxf(m, part_m[w]);
  float o = 0.f, l = 0.f;
#pragma unroll
  for (int w = 0; w < FD_WAVES; ++w) {
    const float scl = __expf(part_m[w] - m);
    o = fmaf(part_o[w][lane], scl, o);
    l = fmaf(part_l[w], scl, l);
  }
  att_h[lane] = (_Float16)(attn16 ? o : o / l);
}
template <int SLOTS, bool MERGE = true, class AfterScores, class Mid>
__device__ __forceinline__ void fd_attend(const half8 (&kr)[SLOTS], half8 (&vr)[SLOTS], const float* q_s, int k_lo, int k_hi,
                                          int attn16, float (*part_o)[64], float* part_m, float* part_l, _Float16* att_h,
                                          AfterScores after_scores, Mid mid) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 7, r = lane >> 3;
  float qv[8];
  {
    const float4 q0 = *reinterpret_cast<const float4*>(q_s + 8 * c);
    const float4 q1 = *reinterpret_cast<const float4*>(q_s + 8 * c + 4);
    qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w;
    qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    if (attn16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = (float)(_Float16)qv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
  }
  float sc[SLOTS];
  float mloc = -1e30f;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    float v = (float)kr[i][0] * qv[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) v = fmaf((float)kr[i][e], qv[e], v);
    v = sum8(v);
    const bool valid = k_lo + 8 * i + r < k_hi;
    sc[i] = valid ? v : -1e30f;
    mloc = fmaxf(mloc, sc[i]);
  }
  after_scores();
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
  float lsum = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float inv16 = 0.f;
  if (attn16) {                 // the soft-max in full, normalised, THEN rounded (ggml's P.V operand): needs the row's maximum and sum first
    if (lane == 0) part_m[wave] = mloc;
    fd_bar();
    float m = part_m[0];
#pragma unroll
    for (int w = 1; w < FD_WAVES; ++w) m = fmaxf(m, part_m[w]);
    mloc = m;
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) ls += k_lo + 8 * i + r < k_hi ? __expf(sc[i] - mloc) : 0.f;
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) ls += __shfl_xor(ls, off, 64);
    if (lane == 0) part_l[wave] = ls;
    fd_bar();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < FD_WAVES; ++w) l += part_l[w];
    inv16 = 1.f / l;
    fd_bar();
  }
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    float pw = k_lo + 8 * i + r < k_hi ? __expf(sc[i] - mloc) : 0.f;
    if (attn16) pw = (float)(_Float16)(pw * inv16);
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
  mid();
  if (r == 0) {
    *reinterpret_cast<float4*>(&part_o[wave][8 * c]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(&part_o[wave][8 * c + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  if (lane == 0) { part_m[wave] = mloc; part_l[wave] = lsum; }
  if (MERGE) {
    fd_bar();
    if (wave == 0) fd_merge(part_o, part_m, part_l, attn16, att_h);
    fd_bar();
  }
}

// po [RB][D] -> part_out[slice][row0 + r][:]
template <int D, int RB>
__device__ __forceinline__ void fd_store_partial(const float* po, float* part_out, int rows, int row0, int slice) {
  const int tid = threadIdx.x, r = tid / (D / 4), q = tid % (D / 4);
  if (r < RB && row0 + r < rows)
    *reinterpret_cast<float4*>(part_out + ((long)slice * rows + row0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(po + r * D + 4 * q);
}

// ---- self-attention block: one head of RB rows -------------------------------------------------------------------------
// RB rows per workgroup share the head's weights in registers and go through the block TOGETHER: one residual-stream
// assembly, RB LayerNorms on RB waves, the q | k | v products of all rows in one pass of the matrix cores (waves 0 - 11: one
// 16-row weight tile each over all of K; waves 12 - 15 hold the out-projection's tiles), then the attentions' partial passes
// one after the other (a row's keys fill the 16 waves), their merges side by side, and the out-projections together.
template <int D, int NP, int SLOTS, int RB>
__global__ __launch_bounds__(FD_THREADS) void fused_self_kernel(FusedSelfArgs a) {
  constexpr int KS = D / 32;                // k-steps of a K = D product = weight registers (half8) per wave
  constexpr int XLD = D + 8;
  constexpr int OT = D / 64;                // out-projection tiles per wave (waves 12 - 15), two k-steps each
  static_assert(2 * OT == KS, "both roles hold the same number of weight registers");
  __shared__ __attribute__((aligned(16))) float xs[RB * D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];      // LayerNorm gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[FD_NB * XLD];
  __shared__ __attribute__((aligned(16))) float q_s[FD_NB][64];
  __shared__ __attribute__((aligned(16))) _Float16 kv_new[FD_NB][128];    // k | v of this position, as the cache holds them
  __shared__ __attribute__((aligned(16))) float part_o[RB][FD_WAVES][64];
  __shared__ float part_m[RB][FD_WAVES], part_l[RB][FD_WAVES];
  __shared__ __attribute__((aligned(16))) _Float16 att_h[FD_NB][72];
  __shared__ __attribute__((aligned(16))) float po[RB * D];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, row0 = blockIdx.y * RB;
  const int pos = *a.pos_dev;                                          // cache row of this step's token
  const int c8 = lane & 7, r8 = lane >> 3;
  const int n16 = lane & 15, kq = lane >> 4;                           // MFMA roles: row / column l & 15, k-group l >> 4
  // (1) everything this workgroup will read, requested before anything is waited for.  The cache: keys k_off .. pos - 1;
  // slots past them (and the new key, which no cache row holds yet) take the new k | v from LDS below.
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, RB> fin;
  fin.request(a.in, a.rows, row0);
  // LATE: weights + keys + values do not fit the registers together (D = 512 with four key slots: 64 + 32 of 128) -- the
  // cache is then requested behind the q | k | v product and the out-projection's weights behind the attention: two
  // exposed round trips to the L2 for Whisper-base at contexts beyond 256 positions, instead of spills.
  constexpr bool LATE = KS * 4 + RB * SLOTS * 8 >= 96;
  int n_keys[RB], k_lo[RB], k_hi[RB];
  half8 kr[RB][SLOTS], vr[RB][SLOTS];
  auto request_kv = [&] {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int row = min(row0 + rb, a.rows - 1);
      const int k_off = a.key_off ? a.key_off[row] : 0;                // left-padded prompts: the clip's first cache row
      n_keys[rb] = pos + 1 - k_off;                                    // keys the row attends to, its own included (>= 1 in a generated step)
      const int per = (max(n_keys[rb], 1) + FD_WAVES - 1) / FD_WAVES;
      k_lo[rb] = wave * per;
      k_hi[rb] = min(n_keys[rb], k_lo[rb] + per);
      const int k_cached = max(n_keys[rb] - 2, 0);                     // last key that is in the cache (clamp target)
      // a uniform base (scalar registers) + one 32-bit byte offset per lane and slot, shared by the key and its value
      const char* Kb = reinterpret_cast<const char*>(a.kv + (long)row * a.kv_row_stride + (long)k_off * (2 * D) + h * 64);
      unsigned off[SLOTS];
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) off[i] = (unsigned)((min(k_lo[rb] + 8 * i + r8, k_cached) * (2 * D) + 8 * c8) * 2);
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) kr[rb][i] = *reinterpret_cast<const half8*>(Kb + off[i]);
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) vr[rb][i] = *reinterpret_cast<const half8*>(Kb + 2 * D + off[i]);
    }
  };
  if (!LATE) request_kv();
  // the weights, in MFMA operand order straight from memory (lane: row n16 of the tile, 16 bytes at k = 32 s + 8 kq)
  half8 wa[KS];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (wave < 12) {                          // q | k | v tile `wave`: projection wave / 4, rows 16 (wave % 4) .. + 15 of the head
    const int p = wave >> 2, j0 = 16 * (wave & 3);
    const _Float16* wt = a.wqkv + (long)(h * 12 + wave) * KS * 512;     // packed: fused_pack_kernel, kind 0
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) wa[s2] = ldu<half8>(wt, 2u * (unsigned)(512 * s2 + 8 * lane));
    bias4 = *reinterpret_cast<const f32x4*>(a.bqkv + p * D + h * 64 + j0 + 4 * kq);
  }
  // out-projection (waves 12 - 15): tiles OT (wave - 12) .. + OT - 1 of D / 16, K = the head's 64 columns.  Requested here
  // with everything else (LATE: behind the attention).
  constexpr bool LATE_WO = LATE;
  auto request_wo = [&] {
    const _Float16* wt = a.wo + (long)(h * (D / 16) + OT * (wave - 12)) * 2 * 512;      // packed, kind 1: this wave's tiles are adjacent
#pragma unroll
    for (int u = 0; u < 2 * OT; ++u) wa[u] = ldu<half8>(wt, 2u * (unsigned)(512 * u + 8 * lane));
  };
  if (!LATE_WO && wave >= 12) request_wo();
  __builtin_amdgcn_sched_barrier(0);
  // (2) residual stream + LayerNorm of every row
  par.stage(gb);
  fin.template finish<XLD>(a.in, par.b, a.rows, row0, h == 0, xs, gb, xn);
  // (3) q | k | v of the head: D[weight row][step row] on the matrix cores, every row of the step at once
  if (wave < 12) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s2], *reinterpret_cast<const half8*>(xn + n16 * XLD + 32 * s2 + 8 * kq), acc, 0, 0, 0);
    const int p = wave >> 2, j0 = 16 * (wave & 3) + 4 * kq;          // this lane: row n16 of the step, outputs j0 .. j0 + 3 of projection p
    if (p == 0) {
      *reinterpret_cast<f32x4*>(&q_s[n16][j0]) = acc + bias4;
    } else {
      const f32x4 v = acc + bias4;
      *reinterpret_cast<half4v*>(&kv_new[n16][64 * (p - 1) + j0]) = half4v{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
  }
  if (LATE) { __builtin_amdgcn_sched_barrier(0); request_kv(); }
  fd_bar();
  // the cache rows of this position: 8 + 8 sixteen-byte pieces per row
  if (tid < 16 * RB) {
    const int rb = tid >> 4, t = tid & 15;
    if (row0 + rb < a.rows) {
      const half8 v = *reinterpret_cast<const half8*>(&kv_new[rb][8 * t]);
      *reinterpret_cast<half8*>(a.kv + (long)(row0 + rb) * a.kv_row_stride + (long)pos * (2 * D) + (t < 8 ? 0 : D - 64) + h * 64 + 8 * t) = v;
    }
  }
  // (4) attention, row after row: slots at or past the new key read it from LDS (finite values; past the last key the weight is zero)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const half8 kn = *reinterpret_cast<const half8*>(&kv_new[rb][8 * c8]), vn = *reinterpret_cast<const half8*>(&kv_new[rb][64 + 8 * c8]);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const bool from_cache = k_lo[rb] + 8 * i + r8 < n_keys[rb] - 1;
      kr[rb][i] = from_cache ? kr[rb][i] : kn;
      vr[rb][i] = from_cache ? vr[rb][i] : vn;
    }
    fd_attend<SLOTS, false>(kr[rb], vr[rb], q_s[rb], k_lo[rb], k_hi[rb], a.attn16, part_o[rb], part_m[rb], part_l[rb], att_h[rb], [] {}, [] {});
  }
  fd_bar();                                                            // the partials of every row: one barrier, then row r's merge on wave r
  if (wave < RB) fd_merge(part_o[wave], part_m[wave], part_l[wave], a.attn16, att_h[wave]);
  fd_bar();
  // (5) this head's share of the output projection, every row: waves 12 - 15
  if (wave >= 12) {
    if (LATE_WO) request_wo();
#pragma unroll
    for (int j = 0; j < OT; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2 * j + s2], *reinterpret_cast<const half8*>(&att_h[n16][32 * s2 + 8 * kq]), acc, 0, 0, 0);
      if (n16 < RB) *reinterpret_cast<f32x4*>(po + n16 * D + 16 * (OT * (wave - 12) + j) + 4 * kq) = acc;
    }
  }
  fd_bar();
  fd_store_partial<D, RB>(po, a.part_out, a.rows, row0, h);
}

// ---- cross-attention block of one (row, head) ----------------------------------------------------------------------
constexpr int FX_SLOTS = 12;                // 16 waves x 12 slots x 8 keys >= 1536 encoder positions
template <int D, int NP, bool STREAM_KV>
__global__ __launch_bounds__(FD_THREADS) void fused_cross_kernel(FusedCrossArgs a) {
  constexpr int PPL = D / 128;
  __shared__ __attribute__((aligned(16))) float xs[D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];      // LayerNorm gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  __shared__ __attribute__((aligned(16))) float q_s[64];
  __shared__ __attribute__((aligned(16))) float part_o[FD_WAVES][64];
  __shared__ float part_m[FD_WAVES], part_l[FD_WAVES];
  __shared__ __attribute__((aligned(16))) _Float16 att_h[64];
  __shared__ __attribute__((aligned(16))) float po[D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Workgroup -> (row, head).  The `group` rows of a clip (the best-of decoders of a fallback pass) read the SAME keys and
  // values: they are placed on one XCD, one after the other in its dispatch order (workgroups are dealt round-robin over
  // the 8 XCDs: MI355X_MICROARCH.md, observed -- for speed only), so that the clip's K | V of this head comes from HBM once
  // and from that XCD's L2 for the other rows.  group == 1: the plain order (clip-major, head fastest).
  constexpr int H = D / 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int gi = (slot / a.group) * 8 + xcd;                           // which (clip, head)
  if (gi >= (a.rows / a.group) * H) return;                            // the padding of the last round of eight
  const int clip = gi / H, h = gi % H;
  const int row = clip * a.group + slot % a.group;
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, 1> fin;
  fin.request(a.in, a.rows, row);
  const int Tn = a.n_keys;
  const int c8 = lane & 7, r8 = lane >> 3;
  const int per = (Tn + FD_WAVES - 1) / FD_WAVES;
  const int k_lo = wave * per, k_hi = min(Tn, k_lo + per);
  const int k_last = max(k_hi - 1, 0);
  // the clip's keys of this head: [Tn][64] f16, contiguous; values Tn * D halves further on.  Uniform bases (scalar
  // registers) + one 32-bit byte offset per lane and slot, the same for a key and its value (64-bit addresses per
  // slot, kept from the key loads to the value loads, went to scratch)
  const char* Kb = reinterpret_cast<const char*>(a.xkv + (long)clip * a.clip_stride + (long)h * 64 * Tn);
  const char* Vb = Kb + (long)Tn * D * 2;
  unsigned off[FX_SLOTS];
#pragma unroll
  for (int i = 0; i < FX_SLOTS; ++i) off[i] = (unsigned)((min(k_lo + 8 * i + r8, k_last) * 64 + 8 * c8) * 2);
  half8 kr[FX_SLOTS], vr[FX_SLOTS];
#pragma unroll
  for (int i = 0; i < FX_SLOTS; ++i) {
    const half8* p = reinterpret_cast<const half8*>(Kb + off[i]);
    kr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
  }
  const int g = lane >> 4, c = lane & 15;
  const int jrow = 4 * wave + g;
  half8 wq[PPL];
  {
    const _Float16* wh = a.wq + (long)h * 64 * D;                       // uniform
#pragma unroll
    for (int j = 0; j < PPL; ++j) wq[j] = ldu<half8>(wh, 2u * (unsigned)(jrow * D + 8 * c + 128 * j));
  }
  const float bq = a.bq[h * 64 + jrow];
  __builtin_amdgcn_sched_barrier(0);
  par.stage(gb);
  fin.finish(a.in, par.b, a.rows, row, h == 0, xs, gb, xn);
  {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < PPL; ++j) v = dot8(wq[j], *reinterpret_cast<const half8*>(xn + 8 * c + 128 * j), v);
    v = sum16(v) + bq;
    if (c == 0) q_s[jrow] = v;
  }
  fd_bar();
  HeadOut<D> ho;
  fd_attend<FX_SLOTS>(kr, vr, q_s, k_lo, k_hi, a.attn16, part_o, part_m, part_l, att_h,
                      [&] {
                        __builtin_amdgcn_sched_barrier(0);          // (or the scheduler hoists these loads above the scores: 96 registers again)
#pragma unroll
                        for (int i = 0; i < FX_SLOTS; ++i) {
                          const half8* p = reinterpret_cast<const half8*>(Vb + off[i]);
                          vr[i] = STREAM_KV ? __builtin_nontemporal_load(p) : *p;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                      },
                      [&] { ho.request(a.wo, h * 64); });
  ho.finish(att_h, po);
  fd_bar();
  fd_store_partial<D, 1>(po, a.part_out, a.rows, row, h);
}

// ---- MLP block: 128 hidden units of RB rows ---------------------------------------------------------------------------
// RB rows share the chunk's weights in registers and go through the block together (see fused_self_kernel): waves 0 - 7
// hold one 16-row tile of fc1 each over all of K, waves 8 - 15 the D / 16 tiles of fc2 over the chunk's 128 columns.
template <int D, int NP, int RB>
__global__ __launch_bounds__(FD_THREADS) void fused_mlp_kernel(FusedMlpArgs a) {
  constexpr int KS = D / 32;                // k-steps of fc1 = weight registers (half8) per wave
  constexpr int XLD = D + 8;
  constexpr int T2 = D / 128;               // fc2 tiles per wave (waves 8 - 15), four k-steps each
  static_assert(4 * T2 == KS, "both roles hold the same number of weight registers");
  __shared__ __attribute__((aligned(16))) float xs[RB * D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];      // LayerNorm gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[FD_NB * XLD];
  __shared__ __attribute__((aligned(16))) _Float16 hh[FD_NB][136];
  __shared__ __attribute__((aligned(16))) float po[RB * D];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = blockIdx.x, row0 = blockIdx.y * RB;
  const int n16 = lane & 15, kq = lane >> 4;
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, RB> fin;
  fin.request(a.in, a.rows, row0);
  half8 wa[KS];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (wave < 8) {                           // fc1: hidden units 128 ch + 16 wave .. + 15
    const _Float16* wt = a.w1 + (long)(ch * 8 + wave) * KS * 512;       // packed, kind 2
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) wa[s2] = ldu<half8>(wt, 2u * (unsigned)(512 * s2 + 8 * lane));
    bias4 = *reinterpret_cast<const f32x4*>(a.b1 + 128 * ch + 16 * wave + 4 * kq);
  } else {                                  // fc2: output rows 16 (T2 (wave - 8) + j) .. + 15, the chunk's 128 columns
    const _Float16* wt = a.w2 + (long)(ch * (D / 16) + T2 * (wave - 8)) * 4 * 512;      // packed, kind 3
#pragma unroll
    for (int u = 0; u < 4 * T2; ++u) wa[u] = ldu<half8>(wt, 2u * (unsigned)(512 * u + 8 * lane));
  }
  __builtin_amdgcn_sched_barrier(0);
  par.stage(gb);
  fin.template finish<XLD>(a.in, par.b, a.rows, row0, ch == 0, xs, gb, xn);
  if (wave < 8) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s2], *reinterpret_cast<const half8*>(xn + n16 * XLD + 32 * s2 + 8 * kq), acc, 0, 0, 0);
    const f32x4 v = acc + bias4;
    *reinterpret_cast<half4v*>(&hh[n16][16 * wave + 4 * kq]) =
        half4v{(_Float16)gelu_ggml(v[0]), (_Float16)gelu_ggml(v[1]), (_Float16)gelu_ggml(v[2]), (_Float16)gelu_ggml(v[3])};
  }
  fd_bar();
  if (wave >= 8) {
#pragma unroll
    for (int j = 0; j < T2; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[4 * j + s2], *reinterpret_cast<const half8*>(&hh[n16][32 * s2 + 8 * kq]), acc, 0, 0, 0);
      if (n16 < RB) *reinterpret_cast<f32x4*>(po + n16 * D + 16 * (T2 * (wave - 8) + j) + 4 * kq) = acc;
    }
  }
  fd_bar();
  fd_store_partial<D, RB>(po, a.part_out, a.rows, row0, ch);
}

// ---- the step's last block: residual stream complete, final LayerNorm as the f16 row the vocabulary projection reads ----
template <int D, int NP>
__global__ __launch_bounds__(FD_THREADS) void fused_finish_kernel(FusedFinishArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  const int row = blockIdx.x;
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, 1> fin;
  fin.request(a.in, a.rows, row);
  par.stage(gb);
  fin.finish(a.in, par.b, a.rows, row, true, xs, gb, xn);
  if (threadIdx.x < D / 8)
    *reinterpret_cast<half8*>(a.y + (long)row * D + 8 * threadIdx.x) = *reinterpret_cast<const half8*>(xn + 8 * threadIdx.x);
}

// dst[((g * tiles + t) * ksteps + s) * 512 + lane * 8 + e] = W[row0(g, t) + (lane & 15)][col0(g, s) + 8 (lane >> 4) + e]:
// the weights in the order the MFMA A operand wants them -- one wave request = one contiguous KB (the same bytes read row
// by row are sixteen 64-byte segments per request, and a one-row step waited 1.8 us longer per launch for them).
// kind 0: q | k | v of a head (g = head; tile t = projection t / 4, head rows 16 (t % 4) ..; all of K)
// kind 1: out-projection slice of a head (g = head; tile = 16 output rows; K = the head's 64 columns)
// kind 2: fc1 chunk (g = chunk of 128 hidden units; tile = 16 of them; all of K)
// kind 3: fc2 chunk (g = chunk; tile = 16 output rows; K = the chunk's 128 columns)
__global__ __launch_bounds__(256) void fused_pack_kernel(const _Float16* __restrict__ W, _Float16* __restrict__ dst, int D, int kind,
                                                         long pieces) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= pieces) return;
  const int lane = (int)(idx & 63);
  long u = idx >> 6;
  int tiles, ksteps;
  switch (kind) {
    case 0: tiles = 12; ksteps = D / 32; break;
    case 1: tiles = D / 16; ksteps = 2; break;
    case 2: tiles = 8; ksteps = D / 32; break;
    default: tiles = D / 16; ksteps = 4; break;
  }
  const int s2 = (int)(u % ksteps); u /= ksteps;
  const int t = (int)(u % tiles);
  const int g = (int)(u / tiles);
  long row, col, ld;
  switch (kind) {
    case 0: row = (long)(t >> 2) * D + g * 64 + 16 * (t & 3); col = 32 * s2; ld = D; break;
    case 1: row = 16 * t; col = g * 64 + 32 * s2; ld = D; break;
    case 2: row = 128 * g + 16 * t; col = 32 * s2; ld = D; break;
    default: row = 16 * t; col = 128 * g + 32 * s2; ld = 4L * D; break;
  }
  const half8 v = *reinterpret_cast<const half8*>(W + (row + (lane & 15)) * ld + col + 8 * (lane >> 4));
  *reinterpret_cast<half8*>(dst + idx * 8) = v;
}

// Rows per workgroup: one while the launch stays near one workgroup per CU (256 of them), else two (self block; with
// four key slots per wave and row there are no registers for a second row) or two / four / eight (MLP block).
template <int D, int NP>
hipError_t self_launch(const FusedSelfArgs& a, hipStream_t s) {
  const int slots = a.max_keys <= 128 ? 1 : a.max_keys <= 256 ? 2 : 4;
  // rows per workgroup: registers hold RB x slots x 8 of keys and values beside the weights
  const int rb_max = slots == 1 ? (D > 384 ? 2 : 4) : (slots == 2 && D <= 384 ? 2 : 1);
  int rb = 1;
  while (rb < rb_max && (D / 64) * ((a.rows + rb - 1) / rb) > 256 + 64) rb *= 2;
  const dim3 block(FD_THREADS), grid(D / 64, (a.rows + rb - 1) / rb);
#define FD_SELF(SL, RB) hipLaunchKernelGGL((fused_self_kernel<D, NP, SL, RB>), grid, block, 0, s, a)
  if (rb == 4) { if constexpr (D <= 384) FD_SELF(1, 4); }
  else if (rb == 2) { if (slots == 1) FD_SELF(1, 2); else if constexpr (D <= 384) FD_SELF(2, 2); }
  else { if (slots == 1) FD_SELF(1, 1); else if (slots == 2) FD_SELF(2, 1); else FD_SELF(4, 1); }
#undef FD_SELF
  return hipGetLastError();
}
template <int D>
hipError_t mlp_launch(const FusedMlpArgs& a, hipStream_t s) {
  const int per_row = D / 32;
  int rb = 1;
  while (rb < 8 && per_row * ((a.rows + rb - 1) / rb) > 256 + 64) rb *= 2;
  const dim3 block(FD_THREADS), grid(per_row, (a.rows + rb - 1) / rb);
  switch (rb) {
    case 1: hipLaunchKernelGGL((fused_mlp_kernel<D, D / 64, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((fused_mlp_kernel<D, D / 64, 2>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((fused_mlp_kernel<D, D / 64, 4>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((fused_mlp_kernel<D, D / 64, 8>), grid, block, 0, s, a); break;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t fused_pack_weights(const void* W, void* dst, int D, int kind, hipStream_t s) {
  const long elems = kind <= 0 ? 3L * D * D : kind == 1 ? (long)D * D : 4L * D * D;
  const long pieces = elems / 8;
  hipLaunchKernelGGL(fused_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const _Float16*>(W),
                     reinterpret_cast<_Float16*>(dst), D, kind, pieces);
  return hipGetLastError();
}

bool fused_decode_supported(int D, int max_keys, int n_audio_ctx) {
  return (D == 384 || D == 512) && max_keys > 0 && max_keys <= 512 && n_audio_ctx <= FD_WAVES * FX_SLOTS * 8;
}

hipError_t fused_self(const FusedSelfArgs& a, bool first, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  switch (a.D) {
    case 384: return first ? self_launch<384, 0>(a, s) : self_launch<384, 12>(a, s);
    case 512: return first ? self_launch<512, 0>(a, s) : self_launch<512, 16>(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t fused_cross(const FusedCrossArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.group < 1 || a.rows % a.group != 0) return hipErrorInvalidValue;
  const int n_groups = (a.rows / a.group) * (a.D / 64);               // (clip, head) pairs; eight of them per round of the XCDs
  const dim3 grid((unsigned)(8 * a.group * ((n_groups + 7) / 8))), block(FD_THREADS);
  if (a.D == 384) {
    if (a.stream_kv) hipLaunchKernelGGL((fused_cross_kernel<384, 6, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((fused_cross_kernel<384, 6, false>), grid, block, 0, s, a);
  } else if (a.D == 512) {
    if (a.stream_kv) hipLaunchKernelGGL((fused_cross_kernel<512, 8, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((fused_cross_kernel<512, 8, false>), grid, block, 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t fused_mlp(const FusedMlpArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.D == 384) return mlp_launch<384>(a, s);
  if (a.D == 512) return mlp_launch<512>(a, s);
  return hipErrorInvalidValue;
}

hipError_t fused_finish(const FusedFinishArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  const dim3 grid(a.rows), block(FD_THREADS);
  if (a.D == 384) hipLaunchKernelGGL((fused_finish_kernel<384, 12>), grid, block, 0, s, a);
  else if (a.D == 512) hipLaunchKernelGGL((fused_finish_kernel<512, 16>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace crispy
