// asr_common.h -- shared declarations of the log-mel / Whisper path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crispy {

// GELU as ggml's CPU backend computes it [UPSTREAM-RECALL: ggml_vec_gelu_f32 with GGML_GELU_FP16, ggml_gelu_f32] -- precision
// mode 1.  ggml looks the value up in a table indexed by the f16 bit pattern of x, whose entries are
//   f16( 0.5 x (1 + tanhf( sqrt(2 / pi) x (1 + 0.044715 x^2) )) )   evaluated in f32 at the f16 value x,
// with x <= -10 -> 0 and x >= 10 -> x in front of the look-up.  The table is a memo of a pure function, so the same
// thing evaluated on the fly -- round x to f16, the tanh formula in f32, round the result to f16 -- is the table entry,
// except where this tanh (1 - 2 / (e^2u + 1): one v_exp_f32, one v_rcp_f32, ~1e-7 absolute) and libm's tanhf land on
// different sides of an f16 rounding boundary (a 1e-3-relative step for about one value in a thousand).
// 13 plain + 2 transcendental (8-cycle) instructions per value; the exact-erf form of mode 0 (A&S 7.1.26) needs 22 + 2.
__device__ __forceinline__ float gelu_ggml(float x) {
  const float xh = (float)(_Float16)x;
  const float u = (0.79788456080286535588f * xh) * fmaf(0.044715f * xh, xh, 1.0f);
  const float t = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);          // e^(2u); u <= 43 for |x| < 10: no overflow
  const float r = __builtin_amdgcn_rcpf(t + 1.0f);
  const float y = (0.5f * xh) * fmaf(-2.0f, r, 2.0f);                        // 1 + tanh u = 2 - 2 / (e^2u + 1)
  float yh = (float)(_Float16)y;
  yh = x <= -10.0f ? 0.0f : yh;
  return x >= 10.0f ? x : yh;
}

constexpr int MEL_FRAMES = 3000;   // frames the encoder consumes (30 s)
constexpr int MEL_TILE = 64;
constexpr int MEL_TILES = 47;      // 3008 frames computed: the clip maximum also sees the tail frames
constexpr int MEL_RAW_FRAMES = MEL_TILE * MEL_TILES;   // all of them are kept: later windows (seek > 0) read frames >= 3000
constexpr int MEL_BINS = 201;
constexpr int MEL_MAX_MELS = 128;

constexpr int MEL_FW_MAX = 1024;   // non-zero taps of a whole filter bank: the tables of a workgroup live in LDS
struct MelTables {
  float hann[400];
  float2 w400[400];                // exp(-2 pi i k / 400)
  int f_meta[MEL_MAX_MELS];        // per mel bin: first FFT bin | taps << 8 | offset into f_w << 16
  float f_w[MEL_FW_MAX];           // concatenated non-zero filter weights (Whisper's banks: 391 / 394 of them)
};

struct MelArgs {
  const float* pcm;        // [batch][pcm_stride] 16 kHz f32
  long pcm_stride;
  const int* n_samples;    // [batch] (device)
  int n_mel;
  const MelTables* tab;
  float* raw;              // [batch][n_mel][3008] log10 values before normalisation
  int* clip_max;           // [batch] order-preserving int key of the clip maximum
  float* out;              // [batch][n_mel][3000] or null
  float* out_t;            // [batch][3002][n_mel] zero padded frame-major copy or null
  // window mode (mel_window_launch): output k is clip clip_idx[k] starting at mel frame seek[k]; frames past the
  // computed range are pure zero padding (log10 floor), the normalisation uses the clip-wide maximum
  const int* clip_idx;
  const int* seek;
};

hipError_t mel_launch(const MelArgs& a, int batch, hipStream_t s);
hipError_t mel_window_launch(const MelArgs& a, int n, hipStream_t s);

// C[M,N] = A[M,K] . W[N,K]^T (+bias) (GELU) (+residual) (+rowtab[m % period]); all f32, row-major.
// A may be a strided view (lda < K): rows overlap, which is how the two convolutions are expressed.
struct GemmArgs {
  const float* A; long lda; long strideA;     // strideA/strideC/strideR: per-batch element strides (grid z)
  const float* W; long ldw;
  float* C; long ldc; long strideC;
  const float* bias;
  const float* residual; long ldr; long strideR;
  const float* rowtab; int rowtab_period;
  int M, N, K;
  int gelu;                                   // 1: exact erf form (precision mode 0); 2: ggml's form (gelu_ggml, mode 1)
  const int* c_off_dev; long c_off_scale;     // optional: C += (*c_off_dev) * c_off_scale (decoder KV-cache slot)
  // Skinny (decode-step) path only:
  //  * LayerNorm folded into the GEMM: W holds gamma-scaled rows, ln_s[n] = sum_k W[n][k], ln_c[n] = sum_k beta_k
  //    W0[n][k] + bias[n]; A is the *un-normalised* input, whose row mean / rstd are accumulated from the operands the
  //    kernel loads anyway, and C = rstd_m (A.W^T - mean_m ln_s[n]) + ln_c[n].  Saves the LayerNorm launch.
  //  * split output: columns >= n_split go to C2 (+ c_off_dev offset) with leading dimension ldc2 -- q and k|v of a
  //    decoder self-attention block in one launch.
  const float* ln_s; const float* ln_c;
  float* C2; long ldc2; int n_split;
  int w_half;                                 // skinny kernel: W points to an f16 matrix (ldw in halves), A is rounded to f16 (mode 1)
  int c2_half;                                // skinny kernel: C2 is an f16 buffer (the decoder's self K|V cache in precision mode 1)
  // Tiled kernel only: head-major store for the cross K|V projection.  Row m = (clip b, frame t) with hm_rows frames
  // per clip, column n = (K or V, head, dim): element goes to C[((b * 2 + kv) * heads + head) * hm_rows * 64 + t * 64
  // + dim], so that one (clip, head) K or V block is a contiguous [hm_rows][64] run for the decode-step attention.
  int hm_rows, hm_width;                      // hm_rows = 0: plain row-major C;  hm_width = model width (N = 2 * width)
  int tiled;                                  // != 0: run the 128 x 128 tiled kernel even for a decode-step shape
  // Skinny path, resident quantised weights (asr_quant.h): W is null; rows [p * wq_rows, (p + 1) * wq_rows) of the (row-fused)
  // weight come from the ggml blocks of wq[p], K / 32 blocks per row, de-quantised in registers -- f32 (x wq_gamma[k]: the
  // LayerNorm fold W' = W . diag(gamma)) for the f32 matrix cores, f16 when w_half is set.
  const unsigned char* wq[3];
  int wq_type, wq_rows;
  const float* wq_gamma;
};
constexpr int SKINNY_MAX_M = 512;   // decode steps with up to this many clips use the skinny kernel (row blocks of 32)
hipError_t gemm_f32_nt(const GemmArgs& g, int batch, hipStream_t s);
// decode-step projection straight from resident ggml blocks (GemmArgs::wq): de-quantised in registers
bool skinny_q_supported(const GemmArgs& g, int batch);
hipError_t gemm_skinny_q(const GemmArgs& g, hipStream_t s);
hipError_t convert_f32_to_f16(const float* src, void* dst, long n, hipStream_t s);      // whisper_enc_f16.hip
// ---- precision mode 1, activations in f16 between the encoder kernels (whisper_enc_f16.hip) ----
constexpr int ENC_TP = 1504;   // keys per V^T row: 1500 padded to whole 32-key tiles
struct HGemmArgs {
  const _Float16* A; long lda;      // [M][lda] f16 (lda < K: overlapping rows, the convolutions as strided views)
  const _Float16* W; long ldw;      // [N][ldw] f16
  void* C; long ldc;                // EPI_F16: f16 [M][ldc]; EPI_RES / EPI_TAB: f32 [M][ldc]; EPI_VT: f16 V^T [clip][N][ENC_TP]
  long strideA, strideC;            // per-batch element strides (grid z)
  const float* bias;
  const float* residual; long ldr;  // EPI_RES
  const float* rowtab; int rowtab_period;   // EPI_TAB: C = GELU(acc + bias) + rowtab[m % period][n]
  int M, N, K;
  int gelu;                         // EPI_F16
  int vt_T;                         // EPI_VT / EPI_KVH: rows per clip
  int kv_width;                     // EPI_KVH: model width dt (N = 2 dt); element (clip b, frame t; K|V, head, dim) goes to
                                    //   C[((b * 2 + kv) * heads + head) * vt_T * 64 + t * 64 + dim] as f16
  int xcd_swizzle;                  // all column tiles of a row tile on one XCD
  // K in segments (k_seg > 0, a multiple of 32; K = n * k_seg, n <= 3): segment i of a row of A starts a_seg_off[i] elements
  // from the row's start (W stays [N][K] contiguous) -- the resampler's x_hi | x_lo | x_hi against W_hi | W_hi | W_lo.
  int k_seg;
  long a_seg_off[3];
  // HGEMM_F32 with the rows of C in groups (c_group_rows > 0): row m of the product is row m % c_group_rows of group
  // m / c_group_rows, stored at C + group * c_group_stride + row * ldc if row < c_group_valid, dropped otherwise -- the
  // resampler's streams as ONE row dimension (their last window each is not an output block).
  int c_group_rows, c_group_valid;
  long c_group_stride;
};
// HGEMM_F32: C = acc as f32 [M][ldc], ldc even (rows 8-byte aligned); no bias
constexpr int HGEMM_F16 = 0, HGEMM_RES = 1, HGEMM_VT = 2, HGEMM_TAB = 3, HGEMM_KVH = 4, HGEMM_F32 = 5;
hipError_t gemm_hh(const HGemmArgs& g, int epi, int batch, hipStream_t s);
hipError_t layernorm_f16out(const float* x, const float* gamma, const float* beta, void* y, long rows, int D, hipStream_t s);
hipError_t attn_encoder_h(const void* qk, const void* vt, void* out, int B, int T, int D, int heads, hipStream_t s,
                          int norm16 = 0);      // norm16: normalised probabilities rounded to f16 (precision mode 2)
hipError_t convert_rows_f32_to_f16(const float* src, long lds, void* dst, long ldd, int cols, long rows, hipStream_t s);
// ---- precision mode 1, decode step (whisper_dec_f16.hip): vocabulary projection over a packed f16 embedding ----
size_t vocab_f16_packed_bytes(int V, int K);
hipError_t pack_vocab_f16(const float* E, void* dst, int V, int K, hipStream_t s);
// logits[M][ldc] (f32) = x[M][ldx] (f16) . E^T, E packed by pack_vocab_f16; K in {384, 512, 768, 1024, 1280}
hipError_t vocab_f16(const void* x, long ldx, const void* Ep, float* C, long ldc, int M, int N, int K, hipStream_t s);
// the same with the step's final LayerNorm taken in (fused decode path, M <= VOCAB_FUSE_ROWS rows, K = 384 | 512): the rows
// are assembled from the last MLP block's partials (FusedIn, asr_common.h below) and normalised by every workgroup itself
struct FusedIn;
constexpr int VOCAB_FUSE_ROWS = 4;
hipError_t vocab_f16_fused(const FusedIn& in, const void* Ep, float* C, long ldc, int M, int N, int K, hipStream_t s);
hipError_t layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, long rows, int D, hipStream_t s);
hipError_t attn_encoder_f32(const float* qkv, float* out, int B, int T, int D, int heads, hipStream_t s);
// Rows of a decode step: one query row per clip, or -- the batched prompt step -- G consecutive rows per clip (row =
// clip * G + j, G = AttnRows::group): row b reads the K|V of clip b / G.  key_step = 1: row j sees n_keys_base + j keys
// (causal self-attention over the positions the step itself writes).  One workgroup per (row, head) either way; the rows
// of a clip are neighbours in the grid, so the repeats of a clip's cross K|V are served by the L2s / the Infinity Cache.
// (A variant that walks a clip's rows over K|V registers loaded once needs 96 + ~40 registers at 4 waves per SIMD:
// 468 bytes of scratch per lane -- not built.)
// stream_kv = 1: the K|V of this launch will not be read again before it has left the caches (a decode step over many
// clips): request it non-temporally, so that it does not evict the decoder's weights from the L2s.
// key_off (device, per clip, nullable): the clip's keys start at cache row key_off[clip] -- prompts of different lengths
// are left-padded to a common length so that one batch decodes them in lock step (whisper_full's previous-text
// conditioning); the key count shrinks by the same amount, a row with no key left writes zeros.
// attn16 (precision mode 2, f16 K|V only): ggml's rounding points inside the attention -- the query rounded to f16 in front
// of K.q, and the soft-max taken in full, normalised, and THEN rounded to f16 in front of P.V (mul_mat converts its f32
// operand to the type of the f16 cache it multiplies) [UPSTREAM-RECALL]; costs two more hand-offs between the waves of a
// (row, head), since the rounding needs the sum over all keys first.
struct AttnRows { int group = 1, key_step = 0, stream_kv = 0; const int* key_off = nullptr; int attn16 = 0; };
hipError_t attn_decoder_f32(const float* q, long ldq, const float* kv, long kv_batch_stride, long ldkv, long head_stride,
                            long koff, long voff, int n_keys_base, const int* pos_dev, float* out, long ldo, int B, int heads,
                            hipStream_t s, AttnRows rows = AttnRows());
// the same kernel over an f16 K|V buffer (cross-attention in precision mode 1)
hipError_t attn_decoder_kv16(const float* q, long ldq, const void* kv, long kv_batch_stride, long ldkv, long head_stride,
                             long koff, long voff, int n_keys_base, const int* pos_dev, float* out, long ldo, int B, int heads,
                             hipStream_t s, int max_keys = 0,     // max_keys: upper bound of n_keys_base + *pos_dev (0: n_keys_base)
                             AttnRows rows = AttnRows());
// ---- decode-step projections of the catalog widths at a few rows (whisper_dec_gemv.hip): matrix-vector products over dense
// f16 rows or resident ggml blocks, LayerNorm computed in the consumer, every weight byte of a row requested at once ----
constexpr int GEMV_MAX_M = 4;        // rows a workgroup takes through its weight rows together
constexpr int GEMV_MAX_ROWS = 512;   // rows of a step (gridDim.y chunks of GEMV_MAX_M: a row's arithmetic is that of a step of its own)
constexpr int GEMV_QKV = 0;        // LN(x) . [Wq | Wk | Wv]^T + b: q -> out (f32 [M][ldo]); k | v -> the f16 cache row of this position
constexpr int GEMV_RES = 1;        // out = x . W^T + b + res (f32; out may alias res)
constexpr int GEMV_F32 = 2;        // out = LN(x) . W^T + b (f32)
constexpr int GEMV_GELU16 = 3;     // out16 = f16(gelu_ggml(LN(x) . W^T + b))
constexpr int GEMV_RES_MERGE = 4;  // GEMV_RES whose activation is the cross-attention output as gemv_xattn left it: xpart, merged in the prologue
constexpr int XA_PARTS = 4;        // workgroups per (row, head) of gemv_xattn: each attends a quarter of the clip's keys
constexpr int XA_SLOTS = 3;        // x 8 keys per wave, 16 waves: 384 keys per part
constexpr int XA_PART_FLOATS = 66; // a part's partial soft-max: maximum, sum, 64 weighted value sums (unnormalised)
struct GemvArgs {
  const float* x; const _Float16* x16; long ldx;   // activation rows: f32 (x16 null) or f16
  const float *ln_g, *ln_b;                         // LayerNorm of x in front of the product (null: none)
  const _Float16* w16;                              // dense f16 weights [N][K] row-major, or null:
  const unsigned char* wq[3]; int wq_type, wq_rows; //   ggml blocks (asr_quant.h), rows [p wq_rows, (p + 1) wq_rows) from wq[p]
  const float* bias;                                // [N] or null
  float* out; _Float16* out16; long ldo;
  const float* res;                                 // GEMV_RES
  _Float16* kv; long kv_row_stride; int pos; const int* pos_dev;   // GEMV_QKV: cache of this layer, row stride per clip, position
  const float* xpart;                               // GEMV_RES_MERGE: [M][K / 64][XA_PARTS][XA_PART_FLOATS]
  int M, N, K;
};
// cross-attention of q (one gemv_dec launch, GEMV_F32) over a quarter of the clip's keys, one workgroup per (row, head, quarter):
// partial soft-maxes into `part`
struct XattnArgs {
  const float* q; long ldq;                                        // [rows][ldq]
  const _Float16* xkv; long clip_stride; int n_keys, group;        // the layer's cross K | V (f16, head-major), rows per clip
  float* part;
  int rows, D;
};
hipError_t gemv_xattn(const XattnArgs& a, hipStream_t s);
bool gemv_dec_supported(int D, int rows);
hipError_t gemv_dec(const GemvArgs& g, int epi, hipStream_t s);

// ---- fused decode step (whisper_dec_fused.hip): a generated token's layer in three launches -------------------------
// The residual stream entering a block: x = x_in + bias + part[0] + ... + part[n - 1] (n fixed per kernel: heads or 4 D / 128),
// written to x_out by the row's first workgroup; LayerNorm (ln_g, ln_b) of it feeds the block's first product as f16.
struct FusedIn {
  const float* x_in;                 // [rows][D]
  const float* bias;                 // [D] bias of the projection whose partials are added (unused by a layer's first block of step 0)
  const float* part;                 // [n][rows][D]
  float* x_out;                      // [rows][D]
  const float* ln_g; const float* ln_b;
};
struct FusedSelfArgs {
  FusedIn in;
  const _Float16* wqkv; const float* bqkv;     // q | k | v weights PACKED by fused_pack_weights(kind 0); bias [3 D]
  const _Float16* wo;                          // out-projection PACKED (kind 1)
  _Float16* kv; long kv_row_stride;            // self K | V cache of this layer: [rows][n_text_ctx][2 D] f16
  const int* pos_dev;                          // device: cache row of this step's token
  const int* key_off;                          // [rows] nullable: first cache row of the row's clip (left-padded prompts)
  int attn16, max_keys;                        // max_keys: bound of pos + 1 for this call (slot count of the attention)
  float* part_out;                             // [heads][rows][D]
  int rows, D;
};
struct FusedCrossArgs {
  FusedIn in;
  const _Float16* wq; const float* bq;         // [D][D] f16, [D]
  const _Float16* wo;
  const _Float16* xkv; long clip_stride;       // cross K | V of this layer: per clip [K | V][head][n_keys][64] f16
  int n_keys, group;                           // encoder positions; rows per clip (rows of a clip share its K | V)
  int attn16, stream_kv;
  float* part_out;                             // [heads][rows][D]
  int rows, D;
};
struct FusedMlpArgs {
  FusedIn in;
  const _Float16* w1; const float* b1;         // fc1 PACKED (kind 2), bias [4 D]
  const _Float16* w2;                          // fc2 PACKED (kind 3)
  float* part_out;                             // [4 D / 128][rows][D]
  int rows, D;
};
struct FusedFinishArgs { FusedIn in; _Float16* y; int rows, D; };     // in.part: the last layer's MLP partials
bool fused_decode_supported(int D, int max_keys, int n_audio_ctx);
// f16 weights [rows][ld] -> the tile order the fused kernels' matrix-core operands are requested in (same byte count);
// kind 0: fused q | k | v [3 D][D]; 1: attention out-projection [D][D]; 2: fc1 [4 D][D]; 3: fc2 [D][4 D]
hipError_t fused_pack_weights(const void* W, void* dst, int D, int kind, hipStream_t s);
hipError_t fused_self(const FusedSelfArgs& a, bool first_layer, hipStream_t s);   // first_layer: x_in is complete (no partials)
hipError_t fused_cross(const FusedCrossArgs& a, hipStream_t s);
hipError_t fused_mlp(const FusedMlpArgs& a, hipStream_t s);
hipError_t fused_finish(const FusedFinishArgs& a, hipStream_t s);
// x[r][:] = tok_emb[tokens[r]] + pos_emb[pos + r % rows_per_clip] for B rows (rows_per_clip > 1: the batched prompt step)
hipError_t embed_tokens_f32(const int* tokens, const float* tok_emb, const float* pos_emb, int pos, const int* pos_dev,
                            float* x, int B, int D, hipStream_t s, int rows_per_clip = 1, const int* row_off = nullptr);
// 48 -> 16 kHz resampler (rubato FftFixedIn(.., 1024, 1, 1) geometry)
constexpr int RS_FFT_IN = 1026, RS_FFT_OUT = 342, RS_CHUNK = 1024;
constexpr int RS_K = 1040;   // 1026 padded to the GEMM's k granularity
constexpr int RS_N = 684;    // 342 outputs + 342 of overlap per block
hipError_t rs_prep(const float* in, long in_stride, long n_in, float scale, int wav_s16, float* A, int batch,
                   int n_blk, hipStream_t s);
hipError_t rs_ola(const float* Y, float* out, long out_stride, int batch, int n_blk, hipStream_t s);
// The same map on the f16 matrix cores: x = x_hi + x_lo and W = W_hi + W_lo as f16 pairs (24 bits of each between them),
// y = x_hi W_hi + x_lo W_hi + x_hi W_lo with f32 accumulation -- 3 x the products at 16 x the rate.  A row of the GEMM is
// the window [block - 1 | block] of a stream (2 x RS_PITCH halves, rows overlap: lda = RS_PITCH), its 342 outputs are the
// block's output samples with the overlap of the previous block already added.
constexpr int RS_PITCH = 1056;          // a block's 1026 samples padded to whole 32-wide k-blocks
// planes[2][batch][(n_blk + 1) * RS_PITCH] f16: hi | lo, block 0 of every stream zero (the block before the first)
hipError_t rs_prep_split(const float* in, long in_stride, long n_in, float scale, int wav_s16, void* planes, int batch, int n_blk,
                         hipStream_t s);

// Greedy pick under the timestamp rules (whisper.cpp whisper_process_logits / openai ApplyTimestampRules; the
// semantics are restated in oracle/whisper_oracle.py: timestamp_rules).  One decoding window per clip.
constexpr int TS_RULES_WCPP = 0, TS_RULES_OPENAI = 1;
struct TsState {
  int last, prev;      // the two most recent picks of this window (-1: none)
  int n;               // picks so far
  int last_ts;         // most recent timestamp token that moves the monotonic bound (-1: none)
  int done;            // EOT sampled, or (whisper.cpp) a timestamp within delta_min frames of the end of the audio
  int seek, seek_end;  // window start and audio length in mel frames
  int pad;
};
// whisper_full stops a window (and refuses a chunk) this many mel frames from the end of the audio [UPSTREAM-RECALL:
// `delta_min = 10` = 100 ms in whisper.cpp >= 1.7.6; 1 s before]
constexpr int TS_DELTA_MIN = 10;
// A pick kernel that also starts the next decoder step (x != nullptr): counters = {position of the PREVIOUS step,
// index of this pick, ticket}.  Every workgroup reads them when it starts; it embeds its clip's pick at position
// counters[0] + 1 into x; the workgroup that finishes LAST (ticket) stores the new position and pick index -- all others
// have read the old ones by then, and the kernels behind the pick read the new position.  One launch instead of three
// (pick, embedding, counter kernel).
struct StepFuse {
  const float* tok_emb;                // [V][D]
  const float* pos_emb;                // [n_text_ctx][D]
  float* x;                            // [B][D] residual stream of the next step; nullptr: plain pick (step from step_dev)
  int D;
  int* counters;                       // {pos, step, ticket}
  const unsigned char* tok_emb_q;      // resident quantised embedding (asr_quant.h) instead of tok_emb; nullptr: tok_emb
  int tok_emb_ttype;
  const int* row_off;                  // [B] nullable: cache row r of clip b is position r - row_off[b] (left-padded prompts)
};
struct TsPickArgs {
  const float* logits;                 // [B][ld]: rows padded to a multiple of four floats, so that every row has the same
  long ld;                             // 16-byte alignment and the same split over the threads (sums then do not depend on the row)
  const unsigned char* mask;           // [V] suppressed at every position (nullable)
  const unsigned char* mask_first;     // [V] suppressed at the first position (union with mask; nullable)
  TsState* st;                         // [B]
  int V, beg, eot, not_tok, rules, max_initial_ts;
  int* tokens_out;                     // [B] the pick (fed to the next decoder step)
  int* tokens_all;                     // [steps][B]
  int* tids_all;                       // [steps][B] most probable timestamp token at that step
  float* plog_all;                     // [steps][B] log-probability of the pick (whisper_token_data::plog): log-softmax over
                                       // everything allowed before the probability-mass rule (nullable)
  const int* step_dev;
  int* done_count;                     // number of clips that are done
  int delta_min;
  // sampling (whisper_sample_token(best = false), the temperature ladder of whisper_full): logits / *temperature, then
  // std::discrete_distribution over the probabilities with the uniform variate u_all[step][b] drawn on the host
  // (std::mt19937 + generate_canonical<double, 53>).  u_all == nullptr: greedy arg-max, temperature ignored.
  const float* temperature;            // device scalar
  const double* u_all;                 // [steps][B]
  float* x_scratch;                    // sampling: [B][TS_SCRATCH_ROW] floats, the rule-filtered row between the kernel's passes
  StepFuse fuse;
  // beam search (whisper_sample_token_topk [UPSTREAM-RECALL]): n_cand > 0 -- every row that is not done draws n_cand ids
  // from ITS distribution with the variates u_all[step][b][0 .. n_cand) and records them (id, log-probability, most probable
  // timestamp) in cand_*[b][n_cand]; nothing is committed: which candidate a decoder continues with is decided by
  // beam_advance (below), which writes the rows' states before the next step.
  int n_cand;
  int* cand_tok; float* cand_plog; int* cand_tid;
};
constexpr int TS_MAX_CAND = 8;          // = WHISPER_MAX_DECODERS
constexpr int TS_SCRATCH_ROW = 7 * 8192;  // ts_sample_kernel: TS_NB blocks of TS_BLK ids per row
hipError_t ts_pick(const TsPickArgs& a, int B, hipStream_t s);
// Beam search: row r of a self K | V cache [layers][rows][row_bytes] continues the sequence of row parent[r] -- the cache rows
// of the positions generated so far, [counters[4], counters[0]) (read on the device: part of a captured step), of every
// layer's row parent[r] become row r's (rows with parent[r] == r are left alone).  Two launches through `scratch`
// [layers][rows][counters[5] * pos_bytes] (a row may be somebody's parent and somebody else's child).  pos_bytes % 16 == 0.
hipError_t beam_kv_reorder(void* kv, void* scratch, const int* parent_dev, int layers, int rows, long row_bytes, long pos_bytes,
                           const int* counters, hipStream_t s);
// One step of whisper_full's BEAM_SEARCH strategy, decided on the device (one wave per clip) so that the step is captured
// and replayed like a greedy one: the candidates the pick kernel drew for the clip's live decoders (TsPickArgs::n_cand) are
// sorted by the sum of all log-probabilities (ties: decoder, draw) and dealt to the live decoders, skipping repeats of the
// sequence just dealt; a decoder takes its candidate's sequence and bookkeeping (BeamRow, TsState), then completion / failure
// on its new last token.  A step's record (id, timestamp id, log-probability, the row the sequence came from) goes to
// rec_*[step][row]: the host walks the parents back from every row's end (BeamRow::n) once the pass is over.
// eqid: two live decoders of a clip hold the same token sequence exactly when their eqid are equal (all equal before the
// first step; afterwards the first decoder dealt the same (parent class, id) this step) -- whisper_sequence_tokens_equal
// without the sequences.  fuse.x != nullptr: the kernel also embeds every row's next input (its new id, EOT once ended) at
// the step's position and moves the counters on, as a fused pick does.
struct BeamRow {
  double sum_all;                      // sum of the log-probabilities of all its tokens
  int has_ts, failed, completed, seek_delta, result_len;
  int n;                               // tokens in its sequence
  int eqid, pad;
};
struct BeamArgs {
  TsState* st;                         // [rows]
  BeamRow* row;                        // [rows]
  const int* cand_tok; const int* cand_tid; const float* cand_plog;      // [rows][n_cand]
  int n_dec, n_cand, rows, beg, eot, rules, delta_min;
  int* rec_tok; int* rec_tid; float* rec_plog; int* rec_parent;         // [max_new][rows]
  int* parent;                         // [rows] for this step's beam_kv_reorder
  int* feed;                           // [rows] the next decoder step's input ids
  int* done_count;                     // decoders that have ended
  const int* counters;                 // {position of the previous step, index of this step, ticket, -, first generated cache row, max_new}
  StepFuse fuse;
};
hipError_t beam_advance(const BeamArgs& a, int n_clips, hipStream_t s);
// p_out[b] = softmax(logits[b])[token] over the whole, unfiltered row (whisper_full's no_speech_prob)
hipError_t softmax_prob_f32(const float* logits, int V, long ld, int token, float* p_out, int B, hipStream_t s);

hipError_t argmax_f32(const float* logits, const unsigned char* mask, const unsigned char* mask_first,
                      const int* step_dev, int V, long ld, int* tokens_out, int* tokens_all, float* best, int B, hipStream_t s,
                      int eot = -1, int* finished = nullptr, int* done_count = nullptr, const StepFuse* fuse = nullptr);

}  // namespace crispy
