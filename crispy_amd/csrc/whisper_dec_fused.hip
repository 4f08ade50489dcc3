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
//
//   * one workgroup = one ROW (clip), 16 waves; every weight byte the workgroup needs (196 KB at D = 384) is requested in
//     its first instructions, together with the residual stream and the self K | V cache, so the chain LayerNorm ->
//     product -> attention -> product inside a launch waits for memory ONCE.  Everything is a matrix-VECTOR product: lanes
//     share a weight row 16 (K = D) or 8 (K = 64) ways, v_dot2c_f32_f16 on the f16 pairs, f32 sums inside the DPP row.
//     No matrix cores: one row has nothing to tile, and a row's arithmetic is trivially the same in every batch.
//
// Arithmetic = ggml's for these products [UPSTREAM-RECALL: mul_mat converts its f32 operand to the f16 of the weight]:
// LayerNorm in f32, its output rounded to f16 against f16 weights, f32 accumulation; q . k with the f32 query against the f16
// cache (mode 2: the query and the normalised probabilities rounded to f16, AttnRows::attn16); attention output and GELU'd
// hidden units rounded to f16 against f16 weights.  Oracle: oracle/whisper_oracle.py DecoderCache(f16=True, ln16=True).
#include "asr_common.h"

namespace crispy {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

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
  __device__ __forceinline__ void finish(const FusedIn& in, float bias, int rows, int row0, bool writer, float* xs, const float* gb,
                                         _Float16* xn) {
    const int tid = threadIdx.x, slot = tid / D, col = tid % D;
    constexpr int PER = D / 64;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int r = ps * S + slot;
      if (slot < S && r < RB) {
        float x = v[ps];
        if (NP > 0) {
          x += bias;
#pragma unroll
          for (int p = 0; p < NP; ++p) x += pv[ps][p];
        }
        xs[r * D + col] = x;
        if (writer && row0 + r < rows) in.x_out[(long)(row0 + r) * D + col] = x;
      }
    }
    fd_bar();
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < RB) {
      const float* xr = xs + wave * D;
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
      for (int q = 0; q < PER; ++q) xn[wave * D + lane + 64 * q] = (_Float16)((e[q] - mean) * rstd * gb[lane + 64 * q] + gb[D + lane + 64 * q]);
    }
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
template <int SLOTS, class AfterScores, class Mid>
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
  fd_bar();
  if (wave == 0) {
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
  fd_bar();
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
// assembly, RB LayerNorms on RB waves, the q | k | v products of all rows off the same weight registers, then the
// attentions one after the other and the out-projections together.  (A decode step of many rows would otherwise run
// more workgroups than the chip holds at once -- ~100 registers x 1024 threads is a whole CU -- each re-reading the
// weights.)  Every row's arithmetic is that of the RB = 1 form, operation for operation.
template <int D, int NP, int SLOTS, int RB>
__global__ __launch_bounds__(FD_THREADS) void fused_self_kernel(FusedSelfArgs a) {
  constexpr int PPL = D / 128;              // 16-byte pieces per lane of a K = D weight row (16 lanes per row)
  __shared__ __attribute__((aligned(16))) float xs[RB * D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];      // LayerNorm gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[RB * D];
  __shared__ __attribute__((aligned(16))) float q_s[RB][64];
  __shared__ __attribute__((aligned(16))) _Float16 kv_new[RB][128];       // k | v of this position, as the cache holds them
  __shared__ __attribute__((aligned(16))) float part_o[FD_WAVES][64];
  __shared__ float part_m[FD_WAVES], part_l[FD_WAVES];
  __shared__ __attribute__((aligned(16))) _Float16 att_h[RB][64];
  __shared__ __attribute__((aligned(16))) float po[RB * D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, row0 = blockIdx.y * RB;
  const int pos = *a.pos_dev;                                          // cache row of this step's token
  const int c8 = lane & 7, r8 = lane >> 3;
  // (1) everything this workgroup will read, requested before anything is waited for.  The cache: keys k_off .. pos - 1;
  // slots past them (and the new key, which no cache row holds yet) take the new k | v from LDS below.
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, RB> fin;
  fin.request(a.in, a.rows, row0);
  int n_keys[RB], k_lo[RB], k_hi[RB];
  half8 kr[RB][SLOTS], vr[RB][SLOTS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = min(row0 + rb, a.rows - 1);
    const int k_off = a.key_off ? a.key_off[row] : 0;                  // left-padded prompts: the clip's first cache row
    n_keys[rb] = pos + 1 - k_off;                                      // keys the row attends to, its own included (>= 1 in a generated step)
    const int per = (max(n_keys[rb], 1) + FD_WAVES - 1) / FD_WAVES;
    k_lo[rb] = wave * per;
    k_hi[rb] = min(n_keys[rb], k_lo[rb] + per);
    const int k_cached = max(n_keys[rb] - 2, 0);                       // last key that is in the cache (clamp target)
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
  // q | k | v rows of this head: pass p = q, k, v; 4 rows per wave and pass, 16 lanes per row
  const int g = lane >> 4, c = lane & 15;
  const int jrow = 4 * wave + g;                                       // 0 .. 63 inside the head
  half8 w3[3][PPL];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const _Float16* wh = a.wqkv + (long)(p * D + h * 64) * D;          // uniform
#pragma unroll
    for (int j = 0; j < PPL; ++j) w3[p][j] = ldu<half8>(wh, 2u * (unsigned)(jrow * D + 8 * c + 128 * j));
  }
  float b3[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) b3[p] = a.bqkv[p * D + h * 64 + jrow];
  __builtin_amdgcn_sched_barrier(0);
  // (2) residual stream + LayerNorm of every row
  par.stage(gb);
  fin.finish(a.in, par.b, a.rows, row0, h == 0, xs, gb, xn);
  // (3) q | k | v of the head, every row
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    half8 xp[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) xp[j] = *reinterpret_cast<const half8*>(xn + rb * D + 8 * c + 128 * j);
    float acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < PPL; ++j) v = dot8(w3[p][j], xp[j], v);
      acc[p] = sum16(v) + b3[p];
    }
    if (c == 0) {
      q_s[rb][jrow] = acc[0];
      kv_new[rb][jrow] = (_Float16)acc[1];
      kv_new[rb][64 + jrow] = (_Float16)acc[2];
    }
  }
  HeadOut<D> ho;
  ho.request(a.wo, h * 64);                                            // in flight during the attention, in the registers the q | k | v weights leave
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
    fd_attend<SLOTS>(kr[rb], vr[rb], q_s[rb], k_lo[rb], k_hi[rb], a.attn16, part_o, part_m, part_l, att_h[rb], [] {}, [] {});
  }
  // (5) this head's share of the output projection, every row
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) ho.finish(att_h[rb], po + rb * D);
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
// RB rows share the chunk's weights in registers and go through the block together (see fused_self_kernel).
template <int D, int NP, int RB>
__global__ __launch_bounds__(FD_THREADS) void fused_mlp_kernel(FusedMlpArgs a) {
  constexpr int PPL = D / 128;
  constexpr int NP2 = D / 64;               // passes of 64 output rows over fc2's D rows
  __shared__ __attribute__((aligned(16))) float xs[RB * D];
  __shared__ __attribute__((aligned(16))) float gb[2 * D];      // LayerNorm gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[RB * D];
  __shared__ __attribute__((aligned(16))) _Float16 hh[RB][128];
  __shared__ __attribute__((aligned(16))) float po[RB * D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ch = blockIdx.x, row0 = blockIdx.y * RB;
  FdParams<D, NP> par;
  par.request(a.in);
  FdInput<D, NP, RB> fin;
  fin.request(a.in, a.rows, row0);
  const int g = lane >> 4, c = lane & 15;
  const int jrow = 4 * wave + g;
  half8 w1[2][PPL], w2[NP2];
  float b1[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const _Float16* wh = a.w1 + (long)(128 * ch + 64 * p) * D;          // uniform
#pragma unroll
    for (int j = 0; j < PPL; ++j) w1[p][j] = ldu<half8>(wh, 2u * (unsigned)(jrow * D + 8 * c + 128 * j));
    b1[p] = a.b1[128 * ch + 64 * p + jrow];
  }
#pragma unroll
  for (int p = 0; p < NP2; ++p)
    w2[p] = ldu<half8>(a.w2 + 128 * ch, 2u * (unsigned)((64 * p + jrow) * (4 * D) + 8 * c));
  __builtin_amdgcn_sched_barrier(0);
  par.stage(gb);
  fin.finish(a.in, par.b, a.rows, row0, ch == 0, xs, gb, xn);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    half8 xp[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) xp[j] = *reinterpret_cast<const half8*>(xn + rb * D + 8 * c + 128 * j);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < PPL; ++j) v = dot8(w1[p][j], xp[j], v);
      v = sum16(v) + b1[p];
      if (c == 0) hh[rb][64 * p + jrow] = (_Float16)gelu_ggml(v);
    }
  }
  fd_bar();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const half8 hp = *reinterpret_cast<const half8*>(&hh[rb][8 * c]);
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
      const float v = sum16(dot8(w2[p], hp, 0.f));
      if (c == 0) po[rb * D + 64 * p + jrow] = v;
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

// Rows per workgroup: one while the launch stays near one workgroup per CU (256 of them), else two (self block; with
// four key slots per wave and row there are no registers for a second row) or two / four / eight (MLP block).
template <int D, int NP>
hipError_t self_launch(const FusedSelfArgs& a, hipStream_t s) {
  const int slots = a.max_keys <= 128 ? 1 : a.max_keys <= 256 ? 2 : 4;
  const bool two = slots <= 2 && (D / 64) * a.rows > 256 + 64;
  const dim3 block(FD_THREADS), grid(D / 64, two ? (a.rows + 1) / 2 : a.rows);
#define FD_SELF(SL, RB) hipLaunchKernelGGL((fused_self_kernel<D, NP, SL, RB>), grid, block, 0, s, a)
  if (two) { if (slots == 1) FD_SELF(1, 2); else FD_SELF(2, 2); }
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
