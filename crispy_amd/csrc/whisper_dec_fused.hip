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

// a load from a workgroup-uniform base (scalar registers) + a 32-bit byte offset per lane: one address register instead of
// two per request -- these kernels keep up to 60 requests per lane in flight
template <class T> __device__ __forceinline__ T ldu(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// The residual stream entering a block, and its LayerNorm as the f16 vector the products read:
//   xs = x_in + bias + part[0] + part[1] + ... (this order), written to x_out by the row's first workgroup;
//   xn = f16(LayerNorm(xs)) -- one wave, a lane holds columns lane + 64 q, two passes (layernorm_h_kernel's arithmetic).
template <int D, int NP>
struct FdInput {
  float v, gm, bt, b, pv[NP > 0 ? NP : 1];
  // the loads, issued FIRST in a kernel: they return first, and the LayerNorm runs while the weights are still arriving
  __device__ __forceinline__ void request(const FusedIn& in, int rows, int row) {
    const int tid = threadIdx.x;
    if (tid < D) {
      v = ldu<float>(in.x_in + (long)row * D, 4u * tid);
      gm = in.ln_g[tid];
      bt = in.ln_b[tid];
      if (NP > 0) {
        b = in.bias[tid];
        const float* prow = in.part + (long)row * D;
#pragma unroll
        for (int p = 0; p < NP; ++p) pv[p] = ldu<float>(prow, 4u * (unsigned)(p * rows * D + tid));
      }
    }
  }
  __device__ __forceinline__ void finish(const FusedIn& in, int row, bool writer, float* xs, _Float16* xn) {
    const int tid = threadIdx.x;
    constexpr int PER = D / 64;
    float* gb = xs + D;                      // gamma | beta, staged by the threads that hold a column (2 registers, not 2 PER)
    if (tid < D) {
      float x = v;
      if (NP > 0) {
        x += b;
#pragma unroll
        for (int p = 0; p < NP; ++p) x += pv[p];
      }
      xs[tid] = x;
      gb[tid] = gm;
      gb[D + tid] = bt;
      if (writer) in.x_out[(long)row * D + tid] = x;
    }
    __syncthreads();
    if (tid < 64) {
      float e[PER], s = 0.f;
#pragma unroll
      for (int q = 0; q < PER; ++q) { e[q] = xs[tid + 64 * q]; s += e[q]; }
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
      for (int q = 0; q < PER; ++q) xn[tid + 64 * q] = (_Float16)((e[q] - mean) * rstd * gb[tid + 64 * q] + gb[D + tid + 64 * q]);
    }
    __syncthreads();
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
  __device__ __forceinline__ void finish(const _Float16* att_h, float* po) const {
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
    __syncthreads();
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
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < FD_WAVES; ++w) l += part_l[w];
    inv16 = 1.f / l;
    __syncthreads();
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
  __syncthreads();
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
  __syncthreads();
}

template <int D>
__device__ __forceinline__ void fd_store_partial(const float* po, float* part_out, int rows, int row, int slice) {
  const int tid = threadIdx.x;
  if (tid < D / 4)
    *reinterpret_cast<float4*>(part_out + ((long)slice * rows + row) * D + 4 * tid) = *reinterpret_cast<const float4*>(po + 4 * tid);
}

// ---- self-attention block of one (row, head) ---------------------------------------------------------------------
template <int D, int NP, int SLOTS>
__global__ __launch_bounds__(FD_THREADS) void fused_self_kernel(FusedSelfArgs a) {
  constexpr int PPL = D / 128;              // 16-byte pieces per lane of a K = D weight row (16 lanes per row)
  __shared__ __attribute__((aligned(16))) float xs[3 * D];      // residual stream | gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  __shared__ __attribute__((aligned(16))) float q_s[64];
  __shared__ __attribute__((aligned(16))) _Float16 kv_new[128];       // k | v of this position, as the cache holds them
  __shared__ __attribute__((aligned(16))) float part_o[FD_WAVES][64];
  __shared__ float part_m[FD_WAVES], part_l[FD_WAVES];
  __shared__ __attribute__((aligned(16))) _Float16 att_h[64];
  __shared__ __attribute__((aligned(16))) float po[D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, row = blockIdx.y;
  const int pos = *a.pos_dev;                                          // cache row of this step's token
  const int k_off = a.key_off ? a.key_off[row] : 0;                    // left-padded prompts: the clip's first cache row
  const int n_keys = pos + 1 - k_off;                                  // keys the row attends to, its own included (>= 1 in a generated step)
  // (1) everything this workgroup will read, requested before anything is waited for.  The cache: keys k_off .. pos - 1;
  // slots past them (and the new key, which no cache row holds yet) take the new k | v from LDS below.
  FdInput<D, NP> fin;
  fin.request(a.in, a.rows, row);
  _Float16* cache = a.kv + (long)row * a.kv_row_stride;
  const int c8 = lane & 7, r8 = lane >> 3;
  const int per = (max(n_keys, 1) + FD_WAVES - 1) / FD_WAVES;
  const int k_lo = wave * per, k_hi = min(n_keys, k_lo + per);
  const int k_cached = max(n_keys - 2, 0);                             // last key that is in the cache (clamp target)
  half8 kr[SLOTS], vr[SLOTS];
  {
    // a uniform base (scalar registers) + one 32-bit byte offset per lane and slot, shared by the key and its value
    const char* Kb = reinterpret_cast<const char*>(cache + (long)k_off * (2 * D) + h * 64);
    unsigned off[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) off[i] = (unsigned)((min(k_lo + 8 * i + r8, k_cached) * (2 * D) + 8 * c8) * 2);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) kr[i] = *reinterpret_cast<const half8*>(Kb + off[i]);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) vr[i] = *reinterpret_cast<const half8*>(Kb + 2 * D + off[i]);
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
  // (2) residual stream + LayerNorm
  fin.finish(a.in, row, h == 0, xs, xn);
  // (3) q | k | v of the head
  {
    half8 xp[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) xp[j] = *reinterpret_cast<const half8*>(xn + 8 * c + 128 * j);
    float acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < PPL; ++j) v = dot8(w3[p][j], xp[j], v);
      acc[p] = sum16(v) + b3[p];
    }
    if (c == 0) {
      q_s[jrow] = acc[0];
      kv_new[jrow] = (_Float16)acc[1];
      kv_new[64 + jrow] = (_Float16)acc[2];
    }
  }
  HeadOut<D> ho;
  ho.request(a.wo, h * 64);                                            // in flight during the attention
  __syncthreads();
  // the cache row of this position: 8 + 8 sixteen-byte pieces
  if (tid < 16) {
    const half8 v = *reinterpret_cast<const half8*>(kv_new + 8 * tid);
    *reinterpret_cast<half8*>(cache + (long)pos * (2 * D) + (tid < 8 ? 0 : D - 64) + h * 64 + 8 * tid) = v;
  }
  // (4) attention: slots at or past the new key read it from LDS (finite values; past the last key the weight is zero)
  {
    const half8 kn = *reinterpret_cast<const half8*>(kv_new + 8 * c8), vn = *reinterpret_cast<const half8*>(kv_new + 64 + 8 * c8);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const bool from_cache = k_lo + 8 * i + r8 < n_keys - 1;
      kr[i] = from_cache ? kr[i] : kn;
      vr[i] = from_cache ? vr[i] : vn;
    }
  }
  fd_attend<SLOTS>(kr, vr, q_s, k_lo, k_hi, a.attn16, part_o, part_m, part_l, att_h, [] {}, [] {});
  // (5) this head's share of the output projection
  ho.finish(att_h, po);
  __syncthreads();
  fd_store_partial<D>(po, a.part_out, a.rows, row, h);
}

// ---- cross-attention block of one (row, head) ----------------------------------------------------------------------
constexpr int FX_SLOTS = 12;                // 16 waves x 12 slots x 8 keys >= 1536 encoder positions
template <int D, int NP, bool STREAM_KV>
__global__ __launch_bounds__(FD_THREADS) void fused_cross_kernel(FusedCrossArgs a) {
  constexpr int PPL = D / 128;
  __shared__ __attribute__((aligned(16))) float xs[3 * D];      // residual stream | gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  __shared__ __attribute__((aligned(16))) float q_s[64];
  __shared__ __attribute__((aligned(16))) float part_o[FD_WAVES][64];
  __shared__ float part_m[FD_WAVES], part_l[FD_WAVES];
  __shared__ __attribute__((aligned(16))) _Float16 att_h[64];
  __shared__ __attribute__((aligned(16))) float po[D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, row = blockIdx.y;
  FdInput<D, NP> fin;
  fin.request(a.in, a.rows, row);
  const int clip = row / a.group;                                      // rows of one clip (best-of decoders) share its K | V
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
  fin.finish(a.in, row, h == 0, xs, xn);
  {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < PPL; ++j) v = dot8(wq[j], *reinterpret_cast<const half8*>(xn + 8 * c + 128 * j), v);
    v = sum16(v) + bq;
    if (c == 0) q_s[jrow] = v;
  }
  __syncthreads();
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
  __syncthreads();
  fd_store_partial<D>(po, a.part_out, a.rows, row, h);
}

// ---- MLP block: 128 hidden units of one row ------------------------------------------------------------------------
template <int D, int NP>
__global__ __launch_bounds__(FD_THREADS) void fused_mlp_kernel(FusedMlpArgs a) {
  constexpr int PPL = D / 128;
  constexpr int NP2 = D / 64;               // passes of 64 output rows over fc2's D rows
  __shared__ __attribute__((aligned(16))) float xs[3 * D];      // residual stream | gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  __shared__ __attribute__((aligned(16))) _Float16 hh[128];
  __shared__ __attribute__((aligned(16))) float po[D];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ch = blockIdx.x, row = blockIdx.y;
  FdInput<D, NP> fin;
  fin.request(a.in, a.rows, row);
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
  fin.finish(a.in, row, ch == 0, xs, xn);
  {
    half8 xp[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) xp[j] = *reinterpret_cast<const half8*>(xn + 8 * c + 128 * j);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < PPL; ++j) v = dot8(w1[p][j], xp[j], v);
      v = sum16(v) + b1[p];
      if (c == 0) hh[64 * p + jrow] = (_Float16)gelu_ggml(v);
    }
  }
  __syncthreads();
  {
    const half8 hp = *reinterpret_cast<const half8*>(hh + 8 * c);
#pragma unroll
    for (int p = 0; p < NP2; ++p) {
      const float v = sum16(dot8(w2[p], hp, 0.f));
      if (c == 0) po[64 * p + jrow] = v;
    }
  }
  __syncthreads();
  fd_store_partial<D>(po, a.part_out, a.rows, row, ch);
}

// ---- the step's last block: residual stream complete, final LayerNorm as the f16 row the vocabulary projection reads ----
template <int D, int NP>
__global__ __launch_bounds__(512) void fused_finish_kernel(FusedFinishArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[3 * D];      // residual stream | gamma | beta
  __shared__ __attribute__((aligned(16))) _Float16 xn[D];
  const int row = blockIdx.x;
  FdInput<D, NP> fin;
  fin.request(a.in, a.rows, row);
  fin.finish(a.in, row, true, xs, xn);
  if (threadIdx.x < D / 8)
    *reinterpret_cast<half8*>(a.y + (long)row * D + 8 * threadIdx.x) = *reinterpret_cast<const half8*>(xn + 8 * threadIdx.x);
}

template <int D>
hipError_t self_launch(const FusedSelfArgs& a, bool first, hipStream_t s) {
  const dim3 grid(D / 64, a.rows), block(FD_THREADS);
  const int slots = a.max_keys <= 128 ? 1 : a.max_keys <= 256 ? 2 : 4;
#define FD_SELF(NP, SL) hipLaunchKernelGGL((fused_self_kernel<D, NP, SL>), grid, block, 0, s, a)
  if (first) { if (slots == 1) FD_SELF(0, 1); else if (slots == 2) FD_SELF(0, 2); else FD_SELF(0, 4); }
  else { if (slots == 1) FD_SELF(D / 32, 1); else if (slots == 2) FD_SELF(D / 32, 2); else FD_SELF(D / 32, 4); }
#undef FD_SELF
  return hipGetLastError();
}

}  // namespace

bool fused_decode_supported(int D, int max_keys, int n_audio_ctx) {
  return (D == 384 || D == 512) && max_keys > 0 && max_keys <= 512 && n_audio_ctx <= FD_WAVES * FX_SLOTS * 8;
}

hipError_t fused_self(const FusedSelfArgs& a, bool first, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  switch (a.D) {
    case 384: return self_launch<384>(a, first, s);
    case 512: return self_launch<512>(a, first, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t fused_cross(const FusedCrossArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  const dim3 grid(a.D / 64, a.rows), block(FD_THREADS);
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
  const dim3 grid(a.D / 32, a.rows), block(FD_THREADS);
  if (a.D == 384) hipLaunchKernelGGL((fused_mlp_kernel<384, 6>), grid, block, 0, s, a);
  else if (a.D == 512) hipLaunchKernelGGL((fused_mlp_kernel<512, 8>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t fused_finish(const FusedFinishArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  const dim3 grid(a.rows), block(512);
  if (a.D == 384) hipLaunchKernelGGL((fused_finish_kernel<384, 12>), grid, block, 0, s, a);
  else if (a.D == 512) hipLaunchKernelGGL((fused_finish_kernel<512, 16>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace crispy
