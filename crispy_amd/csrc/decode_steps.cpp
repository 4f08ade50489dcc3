// decode_steps.cpp -- the Whisper decoder of a crispy_asr handle: workspaces, the three forms of a step (the fused kernels of
// whisper_dec_fused.hip for tiny / base, the matrix-vector products of whisper_dec_gemv.hip for the catalog widths, the skinny
// GEMMs for everything else and for the multi-position prompt), cross K | V, captured steps replayed four tokens at a time,
// and the passes over one window: plain greedy, greedy / sampling under the timestamp rules, beam search.  The decoder half of
// transcribe_rs::SpeechModel::transcribe (src-tauri/src/managers/transcription.rs:183-185); whisper.cpp's decoder graph and
// sampling [UPSTREAM-RECALL].
#include "whisper_internal.h"

using namespace crispy;
using namespace crispy::asr;

namespace crispy {
namespace asr {

// Row stride of h->d_logits: the vocabulary padded to a multiple of four floats.  n_vocab is odd (51865): with rows V
// apart every clip's row has another 16-byte alignment, the pick kernels split it over their threads differently, and a
// sum over the row (the log-probability of a pick) comes out with other last bits for the same logits -- enough to
// reorder two best-of decoders that sampled the same tokens.
long logits_ld(const crispy_asr* h) { return ((long)h->hp.n_vocab + 3) & ~3L; }

// Decoder workspace for `batch` rows (sequences with a self K|V cache of their own) over `xclips` audio clips (cross K|V;
// 0: one clip per row).  Grows only; growing frees everything and drops the captured steps.
int reserve_dec(crispy_asr* h, int batch, int xclips) {
  if (xclips <= 0) xclips = batch;
  if (batch <= h->dcap_batch && xclips <= h->dcap_xclips) return CRISPY_OK;
  batch = std::max(batch, h->dcap_batch);
  xclips = std::max(xclips, h->dcap_xclips);
  free_dec_ws(h);
  const size_t B = batch, X = xclips, dt = h->hp.n_text_state, L = h->hp.n_text_layer, Tn = h->hp.n_audio_ctx,
               C = h->hp.n_text_ctx;
  // activation rows: one per clip in a generation step, up to SKINNY_MAX_M in a batched prompt step (prefill)
  const size_t R = B > (size_t)SKINNY_MAX_M ? B : (size_t)SKINNY_MAX_M;
  HIP_TRY(hipMalloc(&h->d_xkv, L * X * Tn * 2 * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_selfkv, L * B * C * 2 * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_dx, R * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_dxn, R * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_dq, R * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_datt, R * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_dh, R * 4 * dt * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_logits, B * (size_t)logits_ld(h) * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_best, B * C * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_tok, R * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_tokens_all, B * C * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_counters, 8 * sizeof(int)));      // position, pick index, ticket of the fused pick, spare; beam pass: first generated cache row, max_new
  HIP_TRY(hipMalloc(&h->d_ts_state, B * sizeof(TsState)));
  HIP_TRY(hipMalloc(&h->d_tids_all, B * C * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_done_count, sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_finished, B * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_xkv_h, L * X * Tn * 2 * dt * 2));
  HIP_TRY(hipMalloc(&h->d_plog_all, B * C * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_nosp, B * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_u_all, B * C * sizeof(double)));
  HIP_TRY(hipMalloc(&h->d_ts_x, B * (size_t)TS_SCRATCH_ROW * sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_temperature, sizeof(float)));
  HIP_TRY(hipMalloc(&h->d_row_off, B * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_beam_parent, B * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_beam_row, B * sizeof(BeamRow)));
  HIP_TRY(hipMalloc(&h->d_beam_cand, B * 3 * TS_MAX_CAND * sizeof(int)));
  HIP_TRY(hipMalloc(&h->d_beam_rec_parent, B * C * sizeof(int)));
  if (fused_decode_supported((int)dt, 1, (int)Tn)) {
    for (int i = 0; i < 3; ++i) {
      HIP_TRY(hipMalloc(&h->d_fx[i], B * dt * sizeof(float)));
      HIP_TRY(hipMalloc(&h->d_fpart[i], (i == 2 ? dt / 32 : dt / 64) * B * dt * sizeof(float)));
    }
  }
  if (gemv_dec_supported((int)dt, 1))
    HIP_TRY(hipMalloc(&h->d_gvpart, (size_t)std::min<size_t>(B, GEMV_MAX_ROWS) * (dt / 64) * XA_PARTS * XA_PART_FLOATS * sizeof(float)));
  h->dcap_batch = batch;
  h->dcap_xclips = xclips;
  return CRISPY_OK;
}

// the last block of a decoder step: final LayerNorm and vocabulary projection of h->d_dx into h->d_logits
int decoder_logits(crispy_asr* h, int batch, hipStream_t s, const float* x = nullptr) {
  const int dt = h->hp.n_text_state, V = h->hp.n_vocab;
  if (!x) x = h->d_dx;
  const bool fold = batch <= SKINNY_MAX_M && dt % 128 == 0;
  if (h->enc_precision == 1 && h->tok_emb_hp) {
    // the reference's arithmetic: final LayerNorm in f32, rounded to f16, against the f16 embedding, f32 accumulation
    HIP_TRY(layernorm_f16out(x, h->dec_ln_w, h->dec_ln_b, h->d_dxn, batch, dt, s));
    HIP_TRY(vocab_f16(h->d_dxn, dt, h->tok_emb_hp, h->d_logits, logits_ld(h), batch, V, dt, s));
  } else {
    // Vocabulary projection in f32: LayerNorm launch + the 128 x 128 tiled kernel for every batch size.  (Up to 64 clips a
    // persistent LayerNorm-folded kernel, gemm_vocab_f32_kernel, used to run instead -- ~7 us faster per step, but other
    // arithmetic than the tiled path of larger batches: a clip's logits then depended, in the last bits, on the size of
    // the batch it was decoded in.  Mode 0 is the mode the parity claims are made in; one path keeps "alone = in any
    // batch" exact there too.)
    HIP_TRY(layernorm_f32(x, h->dec_ln_w, h->dec_ln_b, h->d_dxn, batch, dt, s));
    GemmArgs g = gemm(h->d_dxn, dt, h->tok_emb, dt, h->d_logits, logits_ld(h), nullptr, batch, V, dt);
    g.tiled = fold ? 1 : 0;
    HIP_TRY(gemm_f32_nt(g, 1, s));
  }
  return CRISPY_OK;
}

// Rows of the biggest decode step the fused kernels take: every step the folded path can hold (SKINNY_MAX_M).  One
// workgroup per (row group, head) holds a head's weights in its registers -- the right shape while the step is a chain of
// latencies (1 row: 0.108 vs 0.173 ms per token staged; 64 rows: 0.222 vs 0.291) and within a few per cent of 32-row
// matrix-core tiles once the rows fill them (Whisper-tiny 512 rows 1.19 vs 1.17 ms).  Rounds 5 switched to the staged
// kernels above 128 rows; the two forms add a row's partial sums in different orders, so a clip's bits -- at a near tie
// its tokens -- depended on whether its batch had more than 128 rows.  Now ONE form decodes every
// generated token of a dense tiny / base model in modes 1 / 2, whatever the batch: a row decodes to the same bits alone
// and in any batch of up to 512 rows (tests/test_gpu_fused_decode.py, tests/test_gpu_pipeline.py cfg 4 / cfg 5 without
// any path override).
constexpr int FUSED_MAX_ROWS = SKINNY_MAX_M;

// CRISPY_ASR_DECODE=stages (developer knob: `make dev` build only, api_util.h): every decode step as one launch per stage --
// the second implementation of the same arithmetic the fused kernels are tested against (tests/test_gpu_fused_decode.py
// loads libcrispy_hip_dev.so for it).  The forms are NOT bit-identical, so the release library does not read it: nothing in
// a host's environment changes a transcript.  Read at the start of a decode call; a change drops the captured steps.
void choose_decode_path(crispy_asr* h) {
  const char* e = dev_env("CRISPY_ASR_DECODE");
  const bool fused = !(e && std::strcmp(e, "stages") == 0);
  if (fused != h->fused_path) {
    (void)hipStreamSynchronize(h->stream);
    h->drop_graphs();
    h->fused_path = fused;
  }
}

bool fused_step_ok(const crispy_asr* h, int rows) {
  return h->fused_path && h->enc_precision == 1 && !h->resident && h->ln16_ready && h->dec[0].qkv_p && h->tok_emb_hp && h->d_fx[0] &&
         rows <= FUSED_MAX_ROWS && fused_decode_supported(h->hp.n_text_state, h->dec_max_keys, h->hp.n_audio_ctx);
}

// A generated token's decoder step through the fused kernels (whisper_dec_fused.hip): 3 launches per layer + the final
// LayerNorm + the vocabulary projection.  The token's embedding is in h->d_dx (written by the pick that chose it), its
// position in h->d_counters[0]; one row per decoder, `rows / xgroup` clips (rows of a clip share its cross K | V).
int decoder_step_fused(crispy_asr* h, int rows, hipStream_t s) {
  const int dt = h->hp.n_text_state, Tn = h->hp.n_audio_ctx, C = h->hp.n_text_ctx;
  const size_t clips = (size_t)rows, xclips = (size_t)(rows / h->cur_xgroup);
  const int attn16 = h->dec_attn16 ? 1 : 0;
  const int stream_kv = xclips * h->dec.size() * Tn * 2 * dt * 2 > ((size_t)256 << 20) ? 1 : 0;      // see decoder_step
  float *xa = h->d_fx[0], *xb = h->d_fx[1], *xc = h->d_fx[2];
  float *pa = h->d_fpart[0], *pb = h->d_fpart[1], *pc = h->d_fpart[2];
  const float* x_in = h->d_dx;
  const float* prev_bias = nullptr;
  for (size_t l = 0; l < h->dec.size(); ++l) {
    const DecLayer& L = h->dec[l];
    FusedSelfArgs a{};
    a.in = FusedIn{x_in, prev_bias, pc, xa, L.ln1_w, L.ln1_b};
    a.wqkv = reinterpret_cast<const _Float16*>(L.qkv_p); a.bqkv = L.qkv_b;
    a.wo = reinterpret_cast<const _Float16*>(L.out_p);
    a.kv = reinterpret_cast<_Float16*>(h->d_selfkv) + l * clips * C * 2 * dt; a.kv_row_stride = (long)C * 2 * dt;
    a.pos_dev = h->d_counters; a.key_off = h->cur_row_off;
    a.attn16 = attn16; a.max_keys = h->dec_max_keys;
    a.part_out = pa; a.rows = rows; a.D = dt;
    HIP_TRY(fused_self(a, l == 0, s));
    FusedCrossArgs b{};
    b.in = FusedIn{xa, L.out_b, pa, xb, L.lnx_w, L.lnx_b};
    b.wq = reinterpret_cast<const _Float16*>(L.xq_wh); b.bq = L.xq_b;
    b.wo = reinterpret_cast<const _Float16*>(L.xout_wh);
    b.xkv = reinterpret_cast<const _Float16*>(h->d_xkv_h) + l * xclips * Tn * 2 * dt; b.clip_stride = (long)Tn * 2 * dt;
    b.n_keys = Tn; b.group = h->cur_xgroup; b.attn16 = attn16;
    b.stream_kv = h->cur_xgroup > 1 ? 0 : stream_kv;      // the rows of a clip share its K | V through the XCD's L2: plain loads
    b.part_out = pb; b.rows = rows; b.D = dt;
    HIP_TRY(fused_cross(b, s));
    FusedMlpArgs m{};
    m.in = FusedIn{xb, L.xout_b, pb, xc, L.ln2_w, L.ln2_b};
    m.w1 = reinterpret_cast<const _Float16*>(L.fc1_p); m.b1 = L.fc1_b;
    m.w2 = reinterpret_cast<const _Float16*>(L.fc2_p);
    m.part_out = pc; m.rows = rows; m.D = dt;
    HIP_TRY(fused_mlp(m, s));
    x_in = xc;
    prev_bias = L.fc2_b;
  }
  FusedFinishArgs f{};
  f.in = FusedIn{xc, prev_bias, pc, xa, h->dec_ln_w, h->dec_ln_b};
  if (rows <= VOCAB_FUSE_ROWS) {       // a few rows: the vocabulary projection normalises them itself (one launch fewer in the chain)
    HIP_TRY(vocab_f16_fused(f.in, h->tok_emb_hp, h->d_logits, logits_ld(h), rows, h->hp.n_vocab, dt, s));
    return CRISPY_OK;
  }
  f.y = reinterpret_cast<_Float16*>(h->d_dxn); f.rows = rows; f.D = dt;
  HIP_TRY(fused_finish(f, s));
  HIP_TRY(vocab_f16(h->d_dxn, dt, h->tok_emb_hp, h->d_logits, logits_ld(h), rows, h->hp.n_vocab, dt, s));
  return CRISPY_OK;
}

// the self K | V cache of a decode call over `rows` rows holds halves (mode 1, folded path) or floats
bool self_kv_half(const crispy_asr* h, int rows) {
  return rows <= SKINNY_MAX_M && h->hp.n_text_state % 128 == 0 && h->enc_precision == 1 && h->dec_max_keys > 0 && h->dec_max_keys <= 512;
}

// A generated-token step of a catalog-width model (768 / 1024 / 1280) in precision mode 1, at ANY row count: the projections as
// matrix-vector products with the LayerNorm computed in the consumer (whisper_dec_gemv.hip) -- 8 launches per layer
// instead of 11, every one at the ~5 us of a dependent launch, N / 8 workgroups instead of N / 32.  Every row count, because a row's arithmetic there is the
// arithmetic of the row decoded alone: one clip, one answer whatever the batch (a form that switched to the skinny kernels
// above four rows would add a row's partial sums in another order -- the hazard round 5 had at 128 rows for the small models).
// Dense f16 copies or resident blocks of ONE ggml type per projection; anything else (a mixed file's dense tensors, precision
// mode 0, the multi-position prompt -- whose form does not depend on the batch either) stays on the skinny kernels.
// CRISPY_ASR_GEMV=0 (developer build) turns it off for the A/B.
bool gemv_ref_ok(const QRef& r) {
  if (r.n <= 0) return false;
  const int tt = r.t[0]->ttype;
  if (tt != QT_Q4_0 && tt != QT_Q4_1 && tt != QT_Q5_0 && tt != QT_Q5_1 && tt != QT_Q8_0) return false;
  for (int i = 1; i < r.n; ++i)
    if (r.t[i]->ttype != tt || r.t[i]->n != r.t[0]->n || r.t[i]->cols != r.t[0]->cols) return false;
  return true;
}
bool gemv_step_ok(const crispy_asr* h, int rows) {
  const char* e = dev_env("CRISPY_ASR_GEMV");       // read per call: a test flips it inside one process (the captured steps are keyed by
  const bool off = e && e[0] == '0';                // the handle, and the two arms of the test use two handles)
  if (off || h->enc_precision != 1 || !h->dec_ln16 || !gemv_dec_supported(h->hp.n_text_state, rows) || !self_kv_half(h, rows)) return false;
  for (const DecLayer& L : h->dec) {
    if (h->resident) {
      if (!(gemv_ref_ok(L.r_qkv) && L.r_qkv.n == 3 && gemv_ref_ok(L.r_out) && gemv_ref_ok(L.r_xq) && gemv_ref_ok(L.r_xout) &&
            gemv_ref_ok(L.r_fc1) && gemv_ref_ok(L.r_fc2)))
        return false;
    } else if (!(L.qkv_wh && L.out_wh && L.xq_wh && L.xout_wh && L.fc1_wh && L.fc2_wh)) {
      return false;
    }
  }
  return true;
}

// ---- one decoder step through the un-fused kernels: what is the same for every layer of the step, then one function per block ----
struct StepCtx {
  crispy_asr* h;
  hipStream_t s;
  int clips, P, rows, pos;             // rows = clips x P
  bool dev_pos;
  const int* pos_dev;                  // h->d_counters with a device position, else null
  bool fold;                           // the skinny kernels (<= SKINNY_MAX_M rows): LayerNorm folded in (mode 0) or a launch of its own (modes 1 / 2)
  AttnRows self_rows, cross_rows;
  int xg;                              // sequences (rows with a self K|V cache of their own) per audio clip
  size_t xclips;
  int dt, H, Tn, C;
};

// (un-folded path only) f32 weights of a projection: the dense tensor, or -- resident model -- the blocks de-quantised into the
// scratch slot in front of the product
const float* step_w32(const StepCtx& c, const float* dense, const QRef& r, const float* gamma, int* rc) {
  if (!c.h->resident) return dense;
  const void* o = nullptr;
  const int e = dq(c.h, r, false, gamma, c.s, &o);
  if (e != CRISPY_OK) *rc = e;
  return reinterpret_cast<const float*>(o);
}

// One projection of the folded path.  Dense model: W = the f32 (gamma-folded) tensor or its f16 copy.  Resident model:
// the skinny kernel reads the ggml blocks itself and de-quantises in registers (gemm_skinny_q); shapes it has no form
// for (and dense tensors of a mixed file) go through the scratch slot and the dense kernel -- f32 x gamma for the
// LayerNorm-folded projections (fold_ln's W' = W . diag(gamma), element for element), f16 for the plain ones.
int step_proj(const StepCtx& c, GemmArgs g, const float* dense32, const void* dense16, const QRef& r, const float* gamma, bool half) {
  crispy_asr* h = c.h;
  g.w_half = half ? 1 : 0;
  if (!h->resident) {
    g.W = half ? reinterpret_cast<const float*>(dense16) : dense32;
    HIP_TRY(gemm_f32_nt(g, 1, c.s));
    return CRISPY_OK;
  }
  bool blocks = r.n > 0 && r.t[0]->ttype != QT_F32;
  for (int i = 1; i < r.n; ++i) blocks = blocks && r.t[i]->ttype == r.t[0]->ttype && r.t[i]->n == r.t[0]->n;
  if (blocks && skinny_q_supported(g, 1)) {
    g.W = nullptr;
    for (int i = 0; i < 3; ++i) g.wq[i] = r.t[i < r.n ? i : 0]->d;
    g.wq_type = r.t[0]->ttype;
    g.wq_rows = (int)(r.t[0]->n / (size_t)r.t[0]->cols);
    g.wq_gamma = gamma;
    HIP_TRY(gemm_skinny_q(g, c.s));
    return CRISPY_OK;
  }
  const void* o = nullptr;
  const int e = dq(h, r, half, gamma, c.s, &o);
  if (e != CRISPY_OK) return e;
  g.W = reinterpret_cast<const float*>(o);
  HIP_TRY(gemm_f32_nt(g, 1, c.s));
  return CRISPY_OK;
}

// A layer of a generated-token step of a catalog-width model: matrix-vector products (whisper_dec_gemv.hip), 8 launches
int layer_gemv(StepCtx& c, size_t l) {
  crispy_asr* h = c.h;
  hipStream_t s = c.s;
  const DecLayer& L = h->dec[l];
  const int dt = c.dt, batch = c.rows;
  _Float16* kvh = reinterpret_cast<_Float16*>(h->d_selfkv) + l * (size_t)c.clips * c.C * 2 * dt;
  _Float16* hid = reinterpret_cast<_Float16*>(h->d_dh);                  // GELU'd hidden units as the f16 fc2 multiplies
  auto weights = [&](GemvArgs& a, const void* dense16, const QRef& r) {
    if (!h->resident) { a.w16 = reinterpret_cast<const _Float16*>(dense16); return; }
    for (int i = 0; i < 3; ++i) a.wq[i] = r.t[i < r.n ? i : 0]->d;
    a.wq_type = r.t[0]->ttype;
    a.wq_rows = (int)(r.t[0]->n / (size_t)r.t[0]->cols);
  };
  auto residual_proj = [&](const float* x32, const _Float16* x16, long ldx, const void* dense16, const QRef& r, const float* bias, int K) -> int {
    GemvArgs a{};
    a.x = x32; a.x16 = x16; a.ldx = ldx; weights(a, dense16, r); a.bias = bias;
    a.out = h->d_dx; a.res = h->d_dx; a.ldo = dt; a.M = batch; a.N = dt; a.K = K;
    HIP_TRY(gemv_dec(a, GEMV_RES, s));
    return CRISPY_OK;
  };
  int rc;
  c.self_rows.attn16 = h->dec_attn16 ? 1 : 0;
  // The LayerNorm in front of q | k | v and fc1: up to GEMV_MAX_M rows every workgroup normalises them itself (no launch); beyond,
  // a workgroup would normalise ALL rows, four per pass -- one launch of the same arithmetic (layernorm_h_kernel: the
  // instructions gv_layernorm_wave mirrors; tests/test_gpu_gemv_decode.py: row 0 of 130 == the row alone, bytes) writes
  // them as f16 once and the products read those
  const bool ln_launch = batch > GEMV_MAX_M;
  _Float16* xn16 = reinterpret_cast<_Float16*>(h->d_dxn);
  auto normalised = [&](GemvArgs& a, const float* g_, const float* b_) -> int {
    if (!ln_launch) { a.x = h->d_dx; a.ldx = dt; a.ln_g = g_; a.ln_b = b_; return CRISPY_OK; }
    HIP_TRY(layernorm_f16out(h->d_dx, g_, b_, xn16, batch, dt, s));
    a.x16 = xn16; a.ldx = dt;
    return CRISPY_OK;
  };
  {
    GemvArgs a{};
    if ((rc = normalised(a, L.ln1_w, L.ln1_b)) != CRISPY_OK) return rc;
    weights(a, L.qkv_wh, L.r_qkv); a.bias = L.qkv_b;
    a.out = h->d_dq; a.ldo = dt; a.kv = kvh; a.kv_row_stride = (long)c.C * 2 * dt; a.pos = c.pos; a.pos_dev = c.pos_dev;
    a.M = batch; a.N = 3 * dt; a.K = dt;
    HIP_TRY(gemv_dec(a, GEMV_QKV, s));
  }
  HIP_TRY(attn_decoder_kv16(h->d_dq, dt, kvh, (long)c.C * 2 * dt, 2L * dt, 64, 0, dt, c.dev_pos ? 1 : c.pos + 1, c.pos_dev, h->d_datt, dt,
                            batch, c.H, s, h->dec_max_keys, c.self_rows));
  if ((rc = residual_proj(h->d_datt, nullptr, dt, L.out_wh, L.r_out, L.out_b, dt)) != CRISPY_OK) return rc;
  if (!c.cross_rows.attn16 && h->d_gvpart && c.Tn <= XA_PARTS * 16 * XA_SLOTS * 8) {
    // cross q (one launch), attention over a quarter of the keys per workgroup (heads x XA_PARTS of them per row), the partial
    // soft-maxes merged by the output projection's prologue
    XattnArgs xa{};
    {
      // q of all rows from one launch (projected inside the cross kernel, per (row, head, quarter) workgroup, it was 128 KB of
      // weights each: 13.5 - 16 us per launch at one row, half of a 64-row step)
      GemvArgs a{};
      if ((rc = normalised(a, L.lnx_w, L.lnx_b)) != CRISPY_OK) return rc;
      if (L.xq_wh) a.w16 = reinterpret_cast<const _Float16*>(L.xq_wh); else weights(a, nullptr, L.r_xq);      // (a resident model keeps this one matrix as f16 too: finalize_resident)
      a.bias = L.xq_b; a.out = h->d_dq; a.ldo = dt; a.M = batch; a.N = dt; a.K = dt;
      HIP_TRY(gemv_dec(a, GEMV_F32, s));
      xa.q = h->d_dq; xa.ldq = dt;
    }
    xa.xkv = reinterpret_cast<const _Float16*>(h->d_xkv_h) + l * c.xclips * c.Tn * 2 * dt; xa.clip_stride = (long)c.Tn * 2 * dt;
    xa.n_keys = c.Tn; xa.group = c.xg; xa.part = h->d_gvpart; xa.rows = batch; xa.D = dt;
    HIP_TRY(gemv_xattn(xa, s));
    GemvArgs a{};
    a.xpart = h->d_gvpart; weights(a, L.xout_wh, L.r_xout); a.bias = L.xout_b;
    a.out = h->d_dx; a.res = h->d_dx; a.ldo = dt; a.M = batch; a.N = dt; a.K = dt;
    HIP_TRY(gemv_dec(a, GEMV_RES_MERGE, s));
  } else {
    {
      GemvArgs a{};
      a.x = h->d_dx; a.ldx = dt; a.ln_g = L.lnx_w; a.ln_b = L.lnx_b; weights(a, L.xq_wh, L.r_xq); a.bias = L.xq_b;
      a.out = h->d_dq; a.ldo = dt; a.M = batch; a.N = dt; a.K = dt;
      HIP_TRY(gemv_dec(a, GEMV_F32, s));
    }
    HIP_TRY(attn_decoder_kv16(h->d_dq, dt, reinterpret_cast<const char*>(h->d_xkv_h) + l * c.xclips * c.Tn * 2 * dt * 2,
                              (long)c.Tn * 2 * dt, 64, 64L * c.Tn, 0, (long)c.Tn * dt, c.Tn, nullptr, h->d_datt, dt, batch, c.H, s, 0,
                              c.cross_rows));
    if ((rc = residual_proj(h->d_datt, nullptr, dt, L.xout_wh, L.r_xout, L.xout_b, dt)) != CRISPY_OK) return rc;
  }
  {
    GemvArgs a{};
    if ((rc = normalised(a, L.ln2_w, L.ln2_b)) != CRISPY_OK) return rc;
    weights(a, L.fc1_wh, L.r_fc1); a.bias = L.fc1_b;
    a.out16 = hid; a.ldo = 4L * dt; a.M = batch; a.N = 4 * dt; a.K = dt;
    HIP_TRY(gemv_dec(a, GEMV_GELU16, s));
  }
  return residual_proj(nullptr, hid, 4L * dt, L.fc2_wh, L.r_fc2, L.fc2_b, 4 * dt);
}

// causal self-attention block of a layer on the skinny / tiled kernels: k | v of this position go straight into the cache row
int layer_self_block(StepCtx& c, size_t l) {
  crispy_asr* h = c.h;
  hipStream_t s = c.s;
  const DecLayer& L = h->dec[l];
  const int dt = c.dt, C = c.C, batch = c.rows, pos = c.pos, P = c.P, clips = c.clips;
  const bool dev_pos = c.dev_pos;
  float* selfkv = h->d_selfkv + l * (size_t)clips * C * 2 * dt;
  float* kv_dst = selfkv + (dev_pos ? 0 : (size_t)pos * 2 * dt);
  // mode 1: the self K|V cache is f16, as whisper.cpp's kv_self is (it aliases the f32 cache: every decode call
  // starts with its own prefill); the projection stores halves, the attention requests all its keys up front
  const bool kv16 = self_kv_half(h, batch);
  _Float16* selfkv_h = reinterpret_cast<_Float16*>(h->d_selfkv) + l * (size_t)clips * C * 2 * dt;
  c.self_rows.attn16 = kv16 && h->dec_attn16 ? 1 : 0;
  int qrc = CRISPY_OK;
  if (c.fold) {
    // precision modes 1 / 2: LayerNorm as a launch of its own, its output rounded to f16 on the way into the f16 matrix cores
    // against f16 weights (ggml's mul_mat arithmetic for these products too); mode 0: LayerNorm folded in, f32 operands
    const bool ln16 = h->dec_ln16;
    if (ln16) HIP_TRY(layernorm_f32(h->d_dx, L.ln1_w, L.ln1_b, h->d_dxn, batch, dt, s));
    GemmArgs g = gemm(ln16 ? h->d_dxn : h->d_dx, dt, nullptr, dt, h->d_dq, dt, ln16 ? L.qkv_b : nullptr, batch, 3 * dt, dt);
    if (!ln16) { g.ln_s = L.qkv_ls; g.ln_c = L.qkv_lc; }
    g.C2 = kv_dst; g.ldc2 = (long)C * 2 * dt; g.n_split = dt;
    if (kv16) { g.C2 = reinterpret_cast<float*>(selfkv_h + (dev_pos ? 0 : (size_t)pos * 2 * dt)); g.c2_half = 1; }
    if (dev_pos) { g.c_off_dev = h->d_counters; g.c_off_scale = 2L * dt; }
    // P rows per clip: k | v of row (clip, j) belongs in cache row (clip, pos + j) -- one clip's P rows are adjacent there,
    // but clips are C rows apart.  One clip: the rows land directly (row stride 2 dt).  Several: staged in the MLP's
    // hidden buffer (free until fc1) and scattered by one strided copy.
    const bool stage_kv = P > 1 && clips > 1;
    if (P > 1) { g.ldc2 = 2L * dt; if (stage_kv) g.C2 = h->d_dh; }
    if ((qrc = step_proj(c, g, L.qkv_lw, L.qkv_wh, L.r_qkv, ln16 ? nullptr : L.ln1_w, ln16)) != CRISPY_OK) return qrc;
    if (stage_kv) {
      const size_t esz = kv16 ? 2 : 4;
      void* dst = kv16 ? static_cast<void*>(selfkv_h + (size_t)pos * 2 * dt) : static_cast<void*>(selfkv + (size_t)pos * 2 * dt);
      HIP_TRY(hipMemcpy2DAsync(dst, (size_t)C * 2 * dt * esz, h->d_dh, (size_t)P * 2 * dt * esz, (size_t)P * 2 * dt * esz,
                               (size_t)clips, hipMemcpyDeviceToDevice, s));
    }
  } else {
    HIP_TRY(layernorm_f32(h->d_dx, L.ln1_w, L.ln1_b, h->d_dxn, batch, dt, s));
    const float* qkv_w = step_w32(c, L.qkv_w, L.r_qkv, nullptr, &qrc);
    if (qrc != CRISPY_OK) return qrc;
    HIP_TRY(gemm_f32_nt(gemm(h->d_dxn, dt, qkv_w, dt, h->d_dq, dt, L.qkv_b, batch, dt, dt), 1, s));
    GemmArgs g = gemm(h->d_dxn, dt, qkv_w + (size_t)dt * dt, dt, kv_dst, (long)C * 2 * dt, L.qkv_b + dt, batch, 2 * dt, dt);
    if (dev_pos) { g.c_off_dev = h->d_counters; g.c_off_scale = 2L * dt; }
    HIP_TRY(gemm_f32_nt(g, 1, s));
  }
  if (kv16)
    HIP_TRY(attn_decoder_kv16(h->d_dq, dt, selfkv_h, (long)C * 2 * dt, 2L * dt, 64, 0, dt, dev_pos ? 1 : pos + 1, c.pos_dev,
                              h->d_datt, dt, batch, c.H, s, h->dec_max_keys, c.self_rows));
  else
    HIP_TRY(attn_decoder_f32(h->d_dq, dt, selfkv, (long)C * 2 * dt, 2L * dt, 64, 0, dt, dev_pos ? 1 : pos + 1, c.pos_dev,
                             h->d_datt, dt, batch, c.H, s, c.self_rows));
  // mode 1: the projections that have no LayerNorm in front (attention outputs, the MLP's second GEMM) in ggml's
  // arithmetic -- f16 weights, the f32 activation rounded to f16 on the way into the matrix cores, f32 accumulation
  const bool wh = c.fold && h->enc_precision == 1 && (L.out_wh || h->resident);
  GemmArgs g = gemm(h->d_datt, dt, L.out_w, dt, h->d_dx, dt, L.out_b, batch, dt, dt);
  g.residual = h->d_dx; g.ldr = dt;
  return step_proj(c, g, L.out_w, L.out_wh, L.r_out, nullptr, wh);
}

// cross-attention block over the encoder output (K | V precomputed once per clip), then the MLP
int layer_cross_and_mlp(StepCtx& c, size_t l) {
  crispy_asr* h = c.h;
  hipStream_t s = c.s;
  const DecLayer& L = h->dec[l];
  const int dt = c.dt, Tn = c.Tn, batch = c.rows;
  const float* xkv = h->d_xkv + l * c.xclips * Tn * 2 * dt;
  const bool ln16 = h->dec_ln16;
  const bool wh = c.fold && h->enc_precision == 1 && (L.out_wh || h->resident);
  int qrc = CRISPY_OK;
  if (c.fold) {
    if (ln16) HIP_TRY(layernorm_f32(h->d_dx, L.lnx_w, L.lnx_b, h->d_dxn, batch, dt, s));
    GemmArgs g = gemm(ln16 ? h->d_dxn : h->d_dx, dt, nullptr, dt, h->d_dq, dt, ln16 ? L.xq_b : nullptr, batch, dt, dt);
    if (!ln16) { g.ln_s = L.xq_ls; g.ln_c = L.xq_lc; }
    if ((qrc = step_proj(c, g, L.xq_lw, L.xq_wh, L.r_xq, ln16 ? nullptr : L.lnx_w, ln16)) != CRISPY_OK) return qrc;
  } else {
    HIP_TRY(layernorm_f32(h->d_dx, L.lnx_w, L.lnx_b, h->d_dxn, batch, dt, s));
    const float* xq_w = step_w32(c, L.xq_w, L.r_xq, nullptr, &qrc);
    if (qrc != CRISPY_OK) return qrc;
    HIP_TRY(gemm_f32_nt(gemm(h->d_dxn, dt, xq_w, dt, h->d_dq, dt, L.xq_b, batch, dt, dt), 1, s));
  }
  if (h->enc_precision == 1)
    HIP_TRY(attn_decoder_kv16(h->d_dq, dt, reinterpret_cast<const char*>(h->d_xkv_h) + l * c.xclips * Tn * 2 * dt * 2,
                              (long)Tn * 2 * dt, 64, 64L * Tn, 0, (long)Tn * dt, Tn, nullptr, h->d_datt, dt, batch, c.H, s, 0,
                              c.cross_rows));
  else
    HIP_TRY(attn_decoder_f32(h->d_dq, dt, xkv, (long)Tn * 2 * dt, 64, 64L * Tn, 0, (long)Tn * dt, Tn, nullptr, h->d_datt, dt,
                             batch, c.H, s, c.cross_rows));
  {
    GemmArgs g = gemm(h->d_datt, dt, L.xout_w, dt, h->d_dx, dt, L.xout_b, batch, dt, dt);
    g.residual = h->d_dx; g.ldr = dt;
    if ((qrc = step_proj(c, g, L.xout_w, L.xout_wh, L.r_xout, nullptr, wh)) != CRISPY_OK) return qrc;
  }
  // MLP
  if (c.fold) {
    if (ln16) HIP_TRY(layernorm_f32(h->d_dx, L.ln2_w, L.ln2_b, h->d_dxn, batch, dt, s));
    GemmArgs g = gemm(ln16 ? h->d_dxn : h->d_dx, dt, nullptr, dt, h->d_dh, 4L * dt, ln16 ? L.fc1_b : nullptr, batch, 4 * dt, dt);
    if (!ln16) { g.ln_s = L.fc1_ls; g.ln_c = L.fc1_lc; }
    g.gelu = h->enc_precision == 1 ? 2 : 1;      // mode 1: ggml's GELU (asr_common.h: gelu_ggml)
    if ((qrc = step_proj(c, g, L.fc1_lw, L.fc1_wh, L.r_fc1, ln16 ? nullptr : L.ln2_w, ln16)) != CRISPY_OK) return qrc;
  } else {
    HIP_TRY(layernorm_f32(h->d_dx, L.ln2_w, L.ln2_b, h->d_dxn, batch, dt, s));
    GemmArgs g = gemm(h->d_dxn, dt, step_w32(c, L.fc1_w, L.r_fc1, nullptr, &qrc), dt, h->d_dh, 4L * dt, L.fc1_b, batch, 4 * dt, dt);
    if (qrc != CRISPY_OK) return qrc;
    g.gelu = h->enc_precision == 1 ? 2 : 1;
    HIP_TRY(gemm_f32_nt(g, 1, s));
  }
  GemmArgs g = gemm(h->d_dh, 4L * dt, L.fc2_w, 4L * dt, h->d_dx, dt, L.fc2_b, batch, dt, 4 * dt);
  g.residual = h->d_dx; g.ldr = dt;
  return step_proj(c, g, L.fc2_w, L.fc2_wh, L.r_fc2, nullptr, wh);
}

// one decoder step for all clips: token ids in h->d_tok; leaves logits in h->d_logits.
// dev_pos = false: the position is the host value `pos` (prompt tokens).
// dev_pos = true : the position is read from h->d_counters[0] by the kernels, so the identical launch
//                  sequence can be captured once in a hipGraph and replayed for every generated token.
//
// P > 1 (prefill only: host position, folded path): the step covers P consecutive positions pos .. pos + P - 1 of every
// clip at once -- row = clip * P + j, token ids [batch][P] in h->d_tok.  Every row goes through exactly the arithmetic of
// the one-position step it replaces (the skinny GEMMs split K by K alone; one attention workgroup per (row, head) with the row's
// own key count), so the result is bit-identical to P steps -- at the cost of one.
int decoder_step(crispy_asr* h, int batch, int pos, bool dev_pos, bool want_logits, hipStream_t s, bool embedded = false,
                 int P = 1) {
  const int dt = h->hp.n_text_state, Tn = h->hp.n_audio_ctx;
  const int clips = batch;
  if (P < 1) P = 1;
  // a generated token (its embedding written by the pick, its position on the device): the fused step kernels
  if (P == 1 && dev_pos && embedded && want_logits && fused_step_ok(h, clips)) return decoder_step_fused(h, clips, s);
  StepCtx c{};
  c.h = h; c.s = s; c.clips = clips; c.P = P; c.rows = clips * P; c.pos = pos; c.dev_pos = dev_pos;
  c.pos_dev = dev_pos ? h->d_counters : nullptr;
  c.dt = dt; c.H = h->hp.n_text_head; c.Tn = Tn; c.C = h->hp.n_text_ctx;
  // <= SKINNY_MAX_M rows: the projections run on the skinny kernel (row blocks of 32), which writes q and k|v of the
  // self-attention block from one launch and, in mode 0, folds the preceding LayerNorm in
  c.fold = c.rows <= SKINNY_MAX_M && dt % 128 == 0;
  if (P > 1 && (!c.fold || dev_pos || embedded))
    return fail(CRISPY_ERR_INVALID_ARG, "decoder_step: a multi-position step needs the folded path and a host position");
  c.self_rows.group = P; c.self_rows.key_step = P > 1 ? 1 : 0;
  c.self_rows.key_off = h->cur_row_off;      // left-padded prompts (decode_ts): every clip's keys start at its own cache row
  c.xg = h->cur_xgroup;
  c.xclips = (size_t)(clips / c.xg);
  c.cross_rows.group = P * c.xg;
  // precision mode 2: q and the normalised probabilities rounded to f16 inside the attentions over the f16 caches
  c.cross_rows.attn16 = h->dec_attn16 && h->enc_precision == 1 ? 1 : 0;
  // The cross K|V of all layers and clips against the 256 MB Infinity Cache: while it fits, it is what stays cached from
  // step to step (plain loads: 16 tiny clips = 147 MB, 6.8 ms per call against 7.0 non-temporal); beyond that it is a
  // one-pass stream that only evicts the decoder's weights from the L2s, and is requested non-temporally
  // (AttnRows::stream_kv: 64 tiny clips 11.2 -> 10.3 ms, 256 base clips 46.7 -> 44.2 ms).
  // (not in a multi-position prompt step: the P rows of a clip read the same K|V one after the other, and the repeats are
  // served by the Infinity Cache only if the first read allocates there: 2.06 vs 2.18 ms for the prompt of 128 clips)
  static const bool prompt_nt = dev_env("CRISPY_XKV_PROMPT_NT") != nullptr;      // developer A/B (tools/ab_prompt_nt.sh)
  c.cross_rows.stream_kv = (P == 1 || prompt_nt) && c.xclips * h->dec.size() * Tn * 2 * dt * (h->enc_precision == 1 ? 2 : 4) > ((size_t)256 << 20) ? 1 : 0;
  if (!embedded) {    // (a fused pick has written the residual stream already)
    if (h->resident)
      HIP_TRY(embed_tokens_q(h->d_tok, h->q_tok_emb->d, h->q_tok_emb->ttype, h->dec_pos, pos, c.pos_dev, h->d_dx, c.rows, dt, s, P,
                             h->cur_row_off));
    else
      HIP_TRY(embed_tokens_f32(h->d_tok, h->tok_emb, h->dec_pos, pos, c.pos_dev, h->d_dx, c.rows, dt, s, P, h->cur_row_off));
  }
  // generated tokens only (device position): a prompt position is a row of a multi-position step or -- when batch x P does not
  // divide the prompt -- a single-position step of the SAME skinny kernels, whatever the batch
  const bool use_gemv = P == 1 && dev_pos && gemv_step_ok(h, c.rows);
  for (size_t l = 0; l < h->dec.size(); ++l) {
    int rc;
    if (use_gemv) {
      rc = layer_gemv(c, l);
    } else {
      rc = layer_self_block(c, l);
      if (rc == CRISPY_OK) rc = layer_cross_and_mlp(c, l);
    }
    if (rc != CRISPY_OK) return rc;
  }
  if (want_logits) {
    if (P == 1) return decoder_logits(h, clips, s);
    // the logits of a prompt step are those of its LAST position: gather row (clip, P - 1) of every clip
    HIP_TRY(hipMemcpy2DAsync(h->d_dq, (size_t)dt * 4, h->d_dx + (size_t)(P - 1) * dt, (size_t)P * dt * 4, (size_t)dt * 4,
                             (size_t)clips, hipMemcpyDeviceToDevice, s));
    return decoder_logits(h, clips, s, h->d_dq);
  }
  return CRISPY_OK;
}

// special token ids [UPSTREAM-RECALL, whisper.cpp `whisper_vocab` + the shift applied at load time]: the defaults are the
// English-only layout (n_vocab 51864: eot 50256, sot 50257, translate 50357, transcribe 50358, solm 50359, prev 50360,
// nosp 50361, notimestamps 50362, first timestamp 50363 -- the 99 language slots after sot are kept in the .en vocabulary
// although no prompt uses them); a multilingual vocabulary (n_vocab >= 51865) moves eot / sot up by one and everything
// after the language block by 1 + (number of languages - 99).
Special vocab_specials(int n_vocab) {
  Special sp{};
  sp.multilingual = n_vocab >= 51865;
  const int extra = sp.multilingual ? n_vocab - 51865 : 0;     // large-v3: one more language
  const int eot = sp.multilingual ? 50257 : 50256;
  sp.sot = eot + 1;
  sp.lang0 = sp.sot + 1;
  sp.n_lang = sp.multilingual ? 99 + extra : 0;                // languages a prompt / the detector can name
  sp.n_lang_slots = 99 + extra;                                // ids between sot and translate (always suppressed)
  sp.translate = sp.sot + 100 + extra;
  sp.transcribe = sp.translate + 1;
  sp.solm = sp.translate + 2;
  sp.prev = sp.translate + 3;
  sp.nosp = sp.translate + 4;
  sp.not_ = sp.translate + 5;
  sp.beg = sp.not_ + 1;
  return sp;
}
Special special_tokens(const crispy_asr* h) { return vocab_specials(h->hp.n_vocab); }

// cross K | V of every layer once per window, then the prompt tokens one position at a time (the language
// token may differ per clip); leaves the logits of the last prompt position in h->d_logits
// cross K | V of every layer, once per window (f16 mode: the decode steps stream an f16 copy of it)
int compute_cross_kv(crispy_asr* h, const float* d_enc, int batch, hipStream_t s) {
  const int dt = h->hp.n_text_state, Tn = h->hp.n_audio_ctx;
  if (h->enc_precision == 1 && (h->dec[0].xkv_wh || h->resident)) {
    // The reference's precision: the projection itself on the f16 matrix cores (encoder output and weights rounded
    // to f16, f32 accumulation), written as f16 head-major straight from the epilogue.  (It used to run as an f32 GEMM
    // followed by a conversion pass: 3.9 + 0.8 ms per layer at 256 Whisper-base clips, more than the whole encoder.)
    const long n = (long)batch * Tn * dt;
    _Float16* enc_h = reinterpret_cast<_Float16*>(h->d_xkv);       // the f32 cross K|V buffer is unused in this mode
    HIP_TRY(convert_f32_to_f16(d_enc, enc_h, n, s));
    for (size_t l = 0; l < h->dec.size(); ++l) {
      const void* xkv_wh = h->dec[l].xkv_wh;
      if (h->resident) { const int rq = dq(h, h->dec[l].r_xkv, true, nullptr, s, &xkv_wh); if (rq != CRISPY_OK) return rq; }
      HGemmArgs g{};
      g.A = enc_h; g.lda = dt; g.W = reinterpret_cast<const _Float16*>(xkv_wh); g.ldw = dt;
      g.C = reinterpret_cast<_Float16*>(h->d_xkv_h) + l * (size_t)batch * Tn * 2 * dt;
      g.bias = h->dec[l].xkv_b; g.M = batch * Tn; g.N = 2 * dt; g.K = dt; g.vt_T = Tn; g.kv_width = dt;
      g.xcd_swizzle = h->xcd_swizzle;
      HIP_TRY(gemm_hh(g, HGEMM_KVH, 1, s));
    }
    return CRISPY_OK;
  }
  for (size_t l = 0; l < h->dec.size(); ++l) {
    float* xkv = h->d_xkv + l * (size_t)batch * Tn * 2 * dt;
    // head-major store: per clip [K | V][head][Tn][64], so the decode-step attention streams contiguous runs
    GemmArgs g = gemm(d_enc, dt, h->dec[l].xkv_w, dt, xkv, 2L * dt, h->dec[l].xkv_b, batch * Tn, 2 * dt, dt);
    g.hm_rows = Tn; g.hm_width = dt;
    HIP_TRY(gemm_f32_nt(g, 1, s));
  }
  if (h->enc_precision == 1)
    HIP_TRY(convert_f32_to_f16(h->d_xkv, h->d_xkv_h, (long)h->dec.size() * batch * Tn * 2 * dt, s));
  return CRISPY_OK;
}

// cross K | V of every layer once per window, then the prompts: tok_mat [batch][n_rows] (host) holds every clip's prompt
// RIGHT-aligned -- a clip whose prompt is shorter than n_rows is padded on the left (token 0) with rows it never attends
// to (h->cur_row_off: the padding per clip; nullptr = none).  Leaves the logits of the last prompt position in h->d_logits.
int prefill(crispy_asr* h, const float* d_enc, int batch, const int* tok_mat, int n_rows, hipStream_t s, int* pos_out) {
  {
    const int rc = compute_cross_kv(h, d_enc, batch / h->cur_xgroup, s);
    if (rc != CRISPY_OK) return rc;
  }
  // The prompt runs as multi-position steps: P positions of every clip per step (decoder_step, P > 1), as many as the
  // skinny kernels' row range allows -- batch x P <= SKINNY_MAX_M, so a 4-token prompt of up to 128 clips is ONE step
  // instead of four, and a long prompt (previous-text conditioning: up to 228 tokens per clip) takes one step per
  // 512 / batch positions.  Bit-identical to the position-by-position prefill (CRISPY_ASR_PREFILL=seq keeps that one
  // available for the A/B test).
  const char* pf_env = test_env("CRISPY_ASR_PREFILL");      // read per call: the A/B test flips it inside one process
  const bool seq = pf_env && std::strcmp(pf_env, "seq") == 0;
  const bool fold = batch <= SKINNY_MAX_M && h->hp.n_text_state % 128 == 0;
  const int p_max = (!fold || seq) ? 1 : std::max(1, SKINNY_MAX_M / batch);
  std::vector<int> tok;
  int pos = 0;
  while (pos < n_rows) {
    const int P = std::min(p_max, n_rows - pos);
    tok.resize((size_t)batch * P);
    for (int b = 0; b < batch; ++b)
      for (int j = 0; j < P; ++j) tok[(size_t)b * P + j] = tok_mat[(size_t)b * n_rows + pos + j];
    HIP_TRY(hipMemcpyAsync(h->d_tok, tok.data(), sizeof(int) * tok.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // tok is reused by the next iteration
    const int rc = decoder_step(h, batch, pos, false, pos + P == n_rows, s, false, P);
    if (rc != CRISPY_OK) return rc;
    pos += P;
  }
  *pos_out = pos;
  return CRISPY_OK;
}

StepFuse step_fuse(crispy_asr* h) {
  StepFuse f{h->tok_emb, h->dec_pos, h->d_dx, h->hp.n_text_state, h->d_counters, nullptr, 0, h->cur_row_off};
  if (h->resident) { f.tok_emb_q = h->q_tok_emb->d; f.tok_emb_ttype = h->q_tok_emb->ttype; }
  return f;
}

TsPickArgs ts_args(crispy_asr* h, int rules, const unsigned char* mask, const unsigned char* mask_first) {
  const Special sp = special_tokens(h);
  TsPickArgs a{};
  a.logits = h->d_logits;
  a.ld = logits_ld(h);
  a.mask = mask;
  a.mask_first = mask_first;
  a.st = h->d_ts_state;
  a.V = h->hp.n_vocab;
  a.beg = sp.beg;
  a.eot = h->eot;
  a.not_tok = sp.not_;
  a.rules = rules;
  a.max_initial_ts = 50;     // whisper.cpp max_initial_ts = 1.0 s at 0.02 s per timestamp; HF/openai: 50
  a.tokens_out = h->d_tok;
  a.tokens_all = h->d_tokens_all;
  a.tids_all = h->d_tids_all;
  a.plog_all = h->d_plog_all;
  a.step_dev = h->d_counters + 1;
  a.done_count = h->d_done_count;
  a.delta_min = TS_DELTA_MIN;
  a.temperature = h->d_temperature;
  a.u_all = nullptr;
  a.x_scratch = h->d_ts_x;
  return a;
}

// Tokens per graph replay: a replay costs 8 - 16 us of host / dispatch time whatever it holds (MI355X_MICROARCH.md,
// graph-replay-floor; measured here 7.8 us between the last kernel of a step and the first of the next), so a decode
// loop replays FOUR captured steps at a time and the odd ones singly.  The device counters carry the position from step
// to step inside a replay exactly as between replays.
constexpr int kStepsPerReplay = 4;

// The captured step(s) for `key`: `body()` enqueues ONE generated token on h->stream (pick + decoder step).
template <class Body>
int step_graph(crispy_asr* h, crispy_asr::TsKey key, Body body, hipGraphExec_t* out) {
  auto slot = h->ts_graphs.find(key);
  if (slot == h->ts_graphs.end()) {
    if (h->ts_graphs.size() >= 48) {     // a bound, not a policy: nothing real alternates between this many shapes
      // run_steps launches without waiting (it polls every 8 tokens): an exec replayed a moment ago may still be in flight
      HIP_TRY(hipStreamSynchronize(h->stream));
      h->drop_graphs();
    }
    hipStream_t s = h->stream;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = CRISPY_OK;
    for (int i = 0; i < key.steps && rc == CRISPY_OK; ++i) rc = body();
    const hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc != CRISPY_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIP_TRY(ce);
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIP_TRY(ie);
    slot = h->ts_graphs.emplace(key, exec).first;
  }
  *out = slot->second;
  return CRISPY_OK;
}

// Replays `n_steps` generated tokens (kStepsPerReplay at a time, then singly); every 8 tokens it asks whether every row
// is done (h->d_done_count >= rows) and stops early.  Returns the steps run in *steps_run.
template <class Body>
int run_steps(crispy_asr* h, crispy_asr::TsKey key, int rows, int n_steps, Body body, int* steps_run) {
  hipStream_t s = h->stream;
  int done = 0, ran = 0;
  while (ran < n_steps) {
    key.steps = n_steps - ran >= kStepsPerReplay ? kStepsPerReplay : 1;
    hipGraphExec_t g = nullptr;
    const int rc = step_graph(h, key, body, &g);
    if (rc != CRISPY_OK) return rc;
    HIP_TRY(hipGraphLaunch(g, s));
    const int before = ran;
    ran += key.steps;
    if (ran / 8 != before / 8) {       // every 8 tokens: has every row ended?  (nothing behind its end is returned)
      HIP_TRY(hipMemcpyAsync(&done, h->d_done_count, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (done >= rows) break;
    }
  }
  *steps_run = ran;
  return CRISPY_OK;
}

// One decoding pass over one window per row under the timestamp rules (oracle/whisper_oracle.py: decode_window /
// decode_temperature).  Every row has its own prompt (previous-text conditioning makes them differ in length: they are
// left-padded to the longest and decoded in lock step, each row attending from its own first cache row on -- the
// arithmetic of the row decoded alone, bit for bit).  u == nullptr: greedy arg-max.  u [max_new][rows] (host): the
// sampling pass of the temperature ladder at `temperature` > 0, one uniform variate per (step, row).
// tokens_out / tids_out / plog_out: [rows][max_new]; n_out[b] = picks up to and including the one that ended the window;
// nosp_out[b] = softmax of the last prompt position's unfiltered logits at <|nospeech|>.
// xgroup: rows per audio clip -- d_enc holds batch / xgroup encoder outputs, rows [c * xgroup, (c + 1) * xgroup) decode
// clip c (the best-of decoders of a fallback pass: own prompt, own self K|V cache, own variates, ONE cross K|V).
int decode_ts(crispy_asr* h, const float* d_enc, int batch, const std::vector<std::vector<int>>& prompts, int rules,
              const int* seek, const int* seek_end, int max_new, const unsigned char* mask, const unsigned char* mask_first,
              float temperature, const double* u, int* tokens_out, int* tids_out, float* plog_out, float* nosp_out,
              int* n_out, int xgroup) {
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if ((int)prompts.size() != batch) return fail(CRISPY_ERR_INVALID_ARG, "decode: %zu prompts for %d rows", prompts.size(), batch);
  if (xgroup < 1 || batch % xgroup != 0) return fail(CRISPY_ERR_INVALID_ARG, "decode: %d rows are not whole groups of %d", batch, xgroup);
  int n_rows = 0;
  for (const auto& p : prompts) {
    if (p.empty()) return fail(CRISPY_ERR_INVALID_ARG, "decode: empty prompt");
    for (int t : p)
      if (t < 0 || t >= h->hp.n_vocab) return fail(CRISPY_ERR_INVALID_ARG, "decode: prompt token %d out of range", t);
    n_rows = std::max(n_rows, (int)p.size());
  }
  if (n_rows + max_new > h->hp.n_text_ctx)
    return fail(CRISPY_ERR_INVALID_ARG, "decode: %d prompt + %d new tokens exceed n_text_ctx %d", n_rows, max_new, h->hp.n_text_ctx);
  if (u && !(temperature > 0.f)) return fail(CRISPY_ERR_INVALID_ARG, "decode: sampling needs a temperature > 0");
  int rc = reserve_dec(h, batch, batch / xgroup);
  if (rc != CRISPY_OK) return rc;
  choose_decode_path(h);
  h->dec_max_keys = n_rows + max_new;
  std::vector<int> off(batch), tok_mat((size_t)batch * n_rows, 0);
  for (int b = 0; b < batch; ++b) {
    off[b] = n_rows - (int)prompts[b].size();
    std::copy(prompts[b].begin(), prompts[b].end(), tok_mat.begin() + (size_t)b * n_rows + off[b]);
  }
  HIP_TRY(hipMemcpyAsync(h->d_row_off, off.data(), sizeof(int) * batch, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  struct OffGuard { crispy_asr* h; ~OffGuard() { h->cur_row_off = nullptr; h->cur_xgroup = 1; } } guard{h};
  h->cur_row_off = h->d_row_off;
  h->cur_xgroup = xgroup;
  int pos = 0;
  rc = prefill(h, d_enc, batch, tok_mat.data(), n_rows, s, &pos);
  if (rc != CRISPY_OK) return rc;
  const Special sp = special_tokens(h);
  HIP_TRY(softmax_prob_f32(h->d_logits, h->hp.n_vocab, logits_ld(h), sp.nosp, h->d_nosp, batch, s));
  std::vector<TsState> st(batch);
  for (int b = 0; b < batch; ++b) st[b] = TsState{-1, -1, 0, -1, 0, seek ? seek[b] : 0, seek_end ? seek_end[b] : (1 << 30), 0};
  // {position of the previous step, index of the next pick, ticket}: the fused pick of a replay embeds at counters[0] + 1
  const int counters[4] = {pos - 1, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h->d_counters, counters, sizeof(counters), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_ts_state, st.data(), sizeof(TsState) * batch, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->d_done_count, 0, sizeof(int), s));
  HIP_TRY(hipMemcpyAsync(h->d_temperature, &temperature, sizeof(float), hipMemcpyHostToDevice, s));
  if (u) HIP_TRY(hipMemcpyAsync(h->d_u_all, u, sizeof(double) * (size_t)max_new * batch, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  TsPickArgs pa = ts_args(h, rules, mask, mask_first);
  pa.u_all = u ? h->d_u_all : nullptr;
  int steps_run = 1;      // picks made = decoder steps replayed + the final pick
  if (max_new > 1) {
    const crispy_asr::TsKey key{h->dec_max_keys <= 128 ? 0 : h->dec_max_keys <= 256 ? 1 : 2, u ? 1 : 0, batch, xgroup, rules, mask, 1};
    TsPickArgs pf = pa;                     // the pick of a replay also embeds its token and moves the counters on
    pf.fuse = step_fuse(h);
    int ran = 0;
    rc = run_steps(h, key, batch, max_new - 1, [&]() -> int {
      HIP_TRY(ts_pick(pf, batch, s));
      return decoder_step(h, batch, 0, true, true, s, true);
    }, &ran);
    if (rc != CRISPY_OK) return rc;
    steps_run += ran;
  }
  HIP_TRY(ts_pick(pa, batch, s));   // the last pick needs no further decoder step
  std::vector<int> all((size_t)steps_run * batch), tids((size_t)steps_run * batch);
  std::vector<float> plog((size_t)steps_run * batch), nosp(batch);
  HIP_TRY(hipMemcpyAsync(all.data(), h->d_tokens_all, all.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(tids.data(), h->d_tids_all, tids.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(plog.data(), h->d_plog_all, plog.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(nosp.data(), h->d_nosp, nosp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(st.data(), h->d_ts_state, sizeof(TsState) * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int b = 0; b < batch; ++b) {
    const int n = st[b].n < max_new ? st[b].n : max_new;
    for (int i = 0; i < max_new; ++i) {
      tokens_out[(size_t)b * max_new + i] = i < n ? all[(size_t)i * batch + b] : h->eot;
      if (tids_out) tids_out[(size_t)b * max_new + i] = i < n ? tids[(size_t)i * batch + b] : pa.beg;
      if (plog_out) plog_out[(size_t)b * max_new + i] = i < n ? plog[(size_t)i * batch + b] : 0.f;
    }
    if (nosp_out) nosp_out[b] = nosp[b];
    if (n_out) n_out[b] = n;
  }
  return CRISPY_OK;
}

// std::generate_canonical<double, 53>(std::mt19937) as libstdc++ and libc++ compute it: two draws, (x0 + x1 2^32) / 2^64
double canonical(std::mt19937& g) {
  const double x0 = (double)g(), x1 = (double)g();
  const double u = (x0 + x1 * 4294967296.0) / 18446744073709551616.0;
  return u < 1.0 ? u : std::nextafter(1.0, 0.0);
}

// The end of a beam pass: the steps' records [step][row] (id, timestamp id, log-probability, the row the sequence came from)
// walked back from every row's last token -- token i of a row was dealt at step i, to the row the later record names as its
// parent -- and the decoders' generators moved on by what they consumed.
int beam_collect(crispy_asr* h, int rows, int n_dec, int n_cand, int max_new, int steps_run, const Special& sp,
                 const std::vector<std::mt19937*>& rng, int* tokens_out, int* tids_out, float* plog_out, float* nosp_out, int* n_out) {
  hipStream_t s = h->stream;
  std::vector<BeamRow> br((size_t)rows);
  std::vector<int> rec_tok((size_t)steps_run * rows), rec_tid((size_t)steps_run * rows), rec_par((size_t)steps_run * rows);
  std::vector<float> rec_plog((size_t)steps_run * rows), nosp(rows);
  HIP_TRY(hipMemcpyAsync(rec_tok.data(), h->d_tokens_all, rec_tok.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(rec_tid.data(), h->d_tids_all, rec_tid.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(rec_par.data(), h->d_beam_rec_parent, rec_par.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(rec_plog.data(), h->d_plog_all, rec_plog.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(br.data(), h->d_beam_row, sizeof(BeamRow) * rows, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(nosp.data(), h->d_nosp, nosp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int r = 0; r < rows; ++r) {
    const int n = br[r].n;
    if (n < 0 || n > steps_run || n > max_new) return fail(CRISPY_ERR_HIP, "beam decode: row %d ends with %d tokens after %d steps", r, n, steps_run);
    for (int i = 0; i < max_new; ++i) {
      tokens_out[(size_t)r * max_new + i] = h->eot;
      if (tids_out) tids_out[(size_t)r * max_new + i] = sp.beg;
      if (plog_out) plog_out[(size_t)r * max_new + i] = 0.f;
    }
    int row = r;
    for (int i = n - 1; i >= 0; --i) {       // token i was dealt to `row` at step i, from the decoder rec_par names
      const size_t x = (size_t)i * rows + row;
      if (row / n_dec != r / n_dec || rec_tok[x] < 0 || rec_tok[x] >= h->hp.n_vocab)
        return fail(CRISPY_ERR_HIP, "beam decode: the record of row %d, step %d is not a decoder of its clip", r, i);
      tokens_out[(size_t)r * max_new + i] = rec_tok[x];
      if (tids_out) tids_out[(size_t)r * max_new + i] = rec_tid[x];
      if (plog_out) plog_out[(size_t)r * max_new + i] = rec_plog[x];
      row = rec_par[x];
    }
    if (nosp_out) nosp_out[r] = nosp[r];
    if (n_out) n_out[r] = n;
    rng[r]->discard(2ull * (unsigned long long)n_cand * (unsigned long long)n);      // a decoder was live for exactly the steps that dealt it a token
  }
  return CRISPY_OK;
}

// One pass of whisper_full's BEAM_SEARCH strategy over one window per clip [UPSTREAM-RECALL: whisper_full_with_state,
// whisper_sample_token_topk; restated in oracle/whisper_oracle.py: decode_temperature(beam_size=)].  Every clip has n_dec
// decoders (rows [c n_dec, (c + 1) n_dec): beam_size of them at temperature 0, best_of above) over ONE cross K | V.  Per step:
//   * every decoder that is neither completed nor failed DRAWS n_cand ids from its distribution (std::discrete_distribution
//     over the probabilities the rules leave at this temperature, n_cand variates from the decoder's own generator -- the
//     pick kernel in its candidate form) -> candidates (decoder, sequence + id, sum of ALL log-probabilities);
//   * the clip's candidates are sorted by that sum (descending; ties: decoder index) and dealt to the live decoders in
//     order, skipping candidates whose token sequence equals the one just dealt (not at the first step); a decoder takes
//     the candidate's sequence, window state and the self K | V rows of the decoder it came from; completion / failure
//     bookkeeping as in the sampling pass (beam_advance_kernel, one wave per clip; beam_kv_reorder);
//   * the next decoder step feeds every live row its last id.
// All of it on the device: a step is captured and replayed like a greedy one (pick, deal, cache reorder, decoder step), four
// per graph launch, the host asking every 8 tokens whether every decoder has ended.  What the generators would have yielded is
// drawn ahead: rng[r] is the generator of row r's decoder, a decoder consumes n_cand variates per step it is live, so its
// variates of step i are the i-th n_cand of its stream whatever the other decoders do; the generators themselves are
// moved on afterwards by what their decoders consumed.
// Outputs as decode_ts: the sequence every decoder ENDS with (the steps' records walked back through the parents).
int decode_beam(crispy_asr* h, const float* d_enc, int n_clips, int n_dec, int n_cand, const std::vector<std::vector<int>>& clip_prompts,
                int rules, const int* seek, const int* seek_end, int max_new, const unsigned char* mask, const unsigned char* mask_first,
                float temperature, const std::vector<std::mt19937*>& rng, int* tokens_out, int* tids_out, float* plog_out,
                float* nosp_out, int* n_out) {
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int rows = n_clips * n_dec;
  if (n_clips < 1 || n_dec < 1 || n_dec > TS_MAX_CAND || n_cand < 1 || n_cand > TS_MAX_CAND || (int)clip_prompts.size() != n_clips ||
      (int)rng.size() != rows || max_new < 1)
    return fail(CRISPY_ERR_INVALID_ARG, "beam decode: %d clips x %d decoders, %d candidates", n_clips, n_dec, n_cand);
  int n_rows = 0;
  for (const auto& p : clip_prompts) {
    if (p.empty()) return fail(CRISPY_ERR_INVALID_ARG, "beam decode: empty prompt");
    for (int t : p)
      if (t < 0 || t >= h->hp.n_vocab) return fail(CRISPY_ERR_INVALID_ARG, "beam decode: prompt token %d out of range", t);
    n_rows = std::max(n_rows, (int)p.size());
  }
  if (n_rows + max_new > h->hp.n_text_ctx)
    return fail(CRISPY_ERR_INVALID_ARG, "beam decode: %d prompt + %d new tokens exceed n_text_ctx %d", n_rows, max_new, h->hp.n_text_ctx);
  int rc = reserve_dec(h, rows, n_clips);
  if (rc != CRISPY_OK) return rc;
  choose_decode_path(h);
  h->dec_max_keys = n_rows + max_new;
  const int dt = h->hp.n_text_state, C = h->hp.n_text_ctx, L = (int)h->dec.size();
  std::vector<int> off(rows), tok_mat((size_t)rows * n_rows, 0);
  for (int r = 0; r < rows; ++r) {
    const std::vector<int>& p = clip_prompts[r / n_dec];
    off[r] = n_rows - (int)p.size();
    std::copy(p.begin(), p.end(), tok_mat.begin() + (size_t)r * n_rows + off[r]);
  }
  // the bytes of a row's cache the decoders of a clip can differ in (the generated positions), and the variates of the pass:
  // both grow with the pass, and a captured step holds their addresses
  const size_t esz = self_kv_half(h, rows) ? 2 : 4;
  const size_t row_bytes = (size_t)C * 2 * dt * esz, pos_bytes = (size_t)2 * dt * esz;
  const size_t need_kv = (size_t)L * rows * (size_t)max_new * pos_bytes, need_u = (size_t)max_new * rows * n_cand * sizeof(double);
  if (need_kv > h->beam_kv_bytes || need_u > h->beam_u_bytes) {
    HIP_TRY(hipStreamSynchronize(s));
    h->drop_graphs();
    if (need_kv > h->beam_kv_bytes) {
      if (h->d_beam_kv) (void)hipFree(h->d_beam_kv);
      h->d_beam_kv = nullptr; h->beam_kv_bytes = 0;
      HIP_TRY(hipMalloc(&h->d_beam_kv, need_kv));
      h->beam_kv_bytes = need_kv;
    }
    if (need_u > h->beam_u_bytes) {
      if (h->d_beam_u) (void)hipFree(h->d_beam_u);
      h->d_beam_u = nullptr; h->beam_u_bytes = 0;
      HIP_TRY(hipMalloc(&h->d_beam_u, need_u));
      h->beam_u_bytes = need_u;
    }
  }
  HIP_TRY(hipMemcpyAsync(h->d_row_off, off.data(), sizeof(int) * rows, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  struct OffGuard { crispy_asr* h; ~OffGuard() { h->cur_row_off = nullptr; h->cur_xgroup = 1; } } guard{h};
  h->cur_row_off = h->d_row_off;
  h->cur_xgroup = n_dec;
  int pos = 0;
  rc = prefill(h, d_enc, rows, tok_mat.data(), n_rows, s, &pos);
  if (rc != CRISPY_OK) return rc;
  const Special sp = special_tokens(h);
  HIP_TRY(softmax_prob_f32(h->d_logits, h->hp.n_vocab, logits_ld(h), sp.nosp, h->d_nosp, rows, s));
  const float t_eff = temperature > 0.f ? temperature : 1.0f;       // temperature 0: the logits as they are (x / 1)
  // u[step][row][k]: variate step * n_cand + k of the row's generator (a copy draws; the generator is moved on below)
  std::vector<double> u((size_t)max_new * rows * n_cand);
  for (int r = 0; r < rows; ++r) {
    std::mt19937 g = *rng[r];
    for (int i = 0; i < max_new; ++i)
      for (int k = 0; k < n_cand; ++k) u[((size_t)i * rows + r) * n_cand + k] = canonical(g);
  }
  std::vector<TsState> st((size_t)rows);
  std::vector<BeamRow> br((size_t)rows);
  for (int r = 0; r < rows; ++r) {
    const int c = r / n_dec;
    st[r] = TsState{-1, -1, 0, -1, 0, seek ? seek[c] : 0, seek_end ? seek_end[c] : (1 << 30), 0};
    br[r] = BeamRow{0.0, 0, 0, 0, 3000, 0, 0, 0, 0};
  }
  // {position of the previous step, index of the next pick, ticket, spare, first generated cache row, max_new}
  const int counters[6] = {pos - 1, 0, 0, 0, pos, max_new};
  HIP_TRY(hipMemcpyAsync(h->d_counters, counters, sizeof(counters), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_ts_state, st.data(), sizeof(TsState) * rows, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_beam_row, br.data(), sizeof(BeamRow) * rows, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->d_done_count, 0, sizeof(int), s));
  HIP_TRY(hipMemcpyAsync(h->d_temperature, &t_eff, sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_beam_u, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  TsPickArgs pa = ts_args(h, rules, mask, mask_first);
  pa.u_all = h->d_beam_u;
  pa.n_cand = n_cand;
  pa.cand_tok = h->d_beam_cand; pa.cand_tid = h->d_beam_cand + (size_t)h->dcap_batch * TS_MAX_CAND;
  pa.cand_plog = reinterpret_cast<float*>(h->d_beam_cand + 2 * (size_t)h->dcap_batch * TS_MAX_CAND);
  BeamArgs ba{};
  ba.st = h->d_ts_state; ba.row = h->d_beam_row;
  ba.cand_tok = pa.cand_tok; ba.cand_tid = pa.cand_tid; ba.cand_plog = pa.cand_plog;
  ba.n_dec = n_dec; ba.n_cand = n_cand; ba.rows = rows; ba.beg = sp.beg; ba.eot = h->eot; ba.rules = rules; ba.delta_min = TS_DELTA_MIN;
  ba.rec_tok = h->d_tokens_all; ba.rec_tid = h->d_tids_all; ba.rec_plog = h->d_plog_all; ba.rec_parent = h->d_beam_rec_parent;
  ba.parent = h->d_beam_parent; ba.feed = h->d_tok; ba.done_count = h->d_done_count; ba.counters = h->d_counters;
  int steps_run = 1;
  if (max_new > 1) {
    const crispy_asr::TsKey key{h->dec_max_keys <= 128 ? 0 : h->dec_max_keys <= 256 ? 1 : 2, 16 + n_cand, rows, n_dec, rules, mask, 1};
    BeamArgs bf = ba;                       // the deal of a replay also embeds the rows' next inputs and moves the counters on
    bf.fuse = step_fuse(h);
    int ran = 0;
    rc = run_steps(h, key, rows, max_new - 1, [&]() -> int {
      HIP_TRY(ts_pick(pa, rows, s));
      HIP_TRY(beam_advance(bf, n_clips, s));
      HIP_TRY(beam_kv_reorder(h->d_selfkv, h->d_beam_kv, h->d_beam_parent, L, rows, (long)row_bytes, (long)pos_bytes, h->d_counters, s));
      return decoder_step(h, rows, 0, true, true, s, true);
    }, &ran);
    if (rc != CRISPY_OK) return rc;
    steps_run += ran;
  }
  HIP_TRY(ts_pick(pa, rows, s));            // the last step needs no further decoder step
  HIP_TRY(beam_advance(ba, n_clips, s));
  return beam_collect(h, rows, n_dec, n_cand, max_new, steps_run, sp, rng, tokens_out, tids_out, plog_out, nosp_out, n_out);
}

// pick a token from the current logits (step-aware suppression), record it, run the next step on it,
// advance the device counters: the body of one generated token
int generation_body(crispy_asr* h, int batch, hipStream_t s) {
  const StepFuse f = step_fuse(h);         // pick + embedding of the pick + counters in one launch
  HIP_TRY(argmax_f32(h->d_logits, h->d_suppress, h->d_suppress_first, h->d_counters + 1, h->hp.n_vocab, logits_ld(h), h->d_tok,
                     h->d_tokens_all, h->d_best, batch, s, h->eot, h->d_finished, h->d_done_count, &f));
  return decoder_step(h, batch, 0, true, true, s, true);
}


}  // namespace asr
}  // namespace crispy

extern "C" {

int crispy_asr_set_suppress(crispy_asr* h, const int* ids, int n, int first_only) try {
  if (!h || (n > 0 && !ids)) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_suppress: NULL argument");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_set_suppress: model not finalized");
  std::vector<unsigned char> m(h->hp.n_vocab, 0);
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= h->hp.n_vocab)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_suppress: token id %d out of range", ids[i]);
    m[ids[i]] = 1;
  }
  if (h->sup_all.empty()) h->sup_all.assign(h->hp.n_vocab, 0);
  if (h->sup_first.empty()) h->sup_first.assign(h->hp.n_vocab, 0);
  (first_only ? h->sup_first : h->sup_all) = m;
  std::vector<unsigned char> first(h->hp.n_vocab);
  for (int v = 0; v < h->hp.n_vocab; ++v) first[v] = h->sup_all[v] | h->sup_first[v];   // first position: both lists
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpy(h->d_suppress, h->sup_all.data(), h->sup_all.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_suppress_first, first.data(), first.size(), hipMemcpyHostToDevice));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_set_suppress")

int crispy_asr_stage_logits_device(crispy_asr* h, const float* d_x, int batch, float* d_logits) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_stage_logits_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_stage_logits_device: model not finalized");
  if (batch < 0 || batch > SKINNY_MAX_M) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_stage_logits_device: batch %d outside [0, %d]", batch, SKINNY_MAX_M);
  if (batch == 0) return CRISPY_OK;
  if (!d_x || !d_logits) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_stage_logits_device: NULL argument");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  int rc = reserve_dec(h, batch);
  if (rc != CRISPY_OK) return rc;
  const int dt = h->hp.n_text_state, V = h->hp.n_vocab;
  // through the decode step's own buffers, so that the code under test is decoder_step's last block
  HIP_TRY(hipMemcpyAsync(h->d_dx, d_x, sizeof(float) * batch * dt, hipMemcpyDeviceToDevice, s));
  rc = decoder_logits(h, batch, s);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(d_logits, sizeof(float) * (size_t)V, h->d_logits, sizeof(float) * (size_t)logits_ld(h), sizeof(float) * (size_t)V,
                           (size_t)batch, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_stage_logits_device")

int crispy_asr_decode_greedy_device(crispy_asr* h, const float* d_enc, int batch, const int* prompt, int n_prompt,
                                    int max_new, int* tokens_out, int* n_out, float* logits_out) try {
  return crispy_asr_decode_greedy_lang_device(h, d_enc, batch, prompt, n_prompt, nullptr, max_new, tokens_out, n_out,
                                              logits_out);
} CRISPY_CATCH_RET("crispy_asr_decode_greedy_device")

int crispy_asr_decode_greedy_lang_device(crispy_asr* h, const float* d_enc, int batch, const int* prompt, int n_prompt,
                                         const int* lang_tokens, int max_new, int* tokens_out, int* n_out,
                                         float* logits_out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_greedy_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_decode_greedy_device: model not finalized");
  if (batch < 0 || max_new < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_greedy_device: negative size");
  if (batch == 0 || max_new == 0) return CRISPY_OK;
  if (!d_enc || !prompt || n_prompt <= 0 || !tokens_out)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_greedy_device: NULL argument");
  if (n_prompt + max_new > h->hp.n_text_ctx)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_greedy_device: %d prompt + %d new tokens exceed n_text_ctx %d",
                n_prompt, max_new, h->hp.n_text_ctx);
  for (int i = 0; i < n_prompt; ++i)
    if (prompt[i] < 0 || prompt[i] >= h->hp.n_vocab)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_greedy_device: prompt token %d out of range", prompt[i]);
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  int rc = reserve_dec(h, batch);
  if (rc != CRISPY_OK) return rc;
  choose_decode_path(h);
  h->dec_max_keys = n_prompt + max_new;
  const int V = h->hp.n_vocab;
  int pos = 0;
  {
    std::vector<int> tok_mat((size_t)batch * n_prompt);
    for (int b = 0; b < batch; ++b)
      for (int j = 0; j < n_prompt; ++j) {
        const int t = (j == 1 && lang_tokens) ? lang_tokens[b] : prompt[j];        // per-clip language token
        if (t < 0 || t >= h->hp.n_vocab) return fail(CRISPY_ERR_INVALID_ARG, "decode: language token %d out of range", t);
        tok_mat[(size_t)b * n_prompt + j] = t;
      }
    rc = prefill(h, d_enc, batch, tok_mat.data(), n_prompt, s, &pos);
  }
  if (rc != CRISPY_OK) return rc;
  const int counters[4] = {pos - 1, 0, 0, 0};     // {position of the previous step, index of the next pick, ticket} (StepFuse)
  HIP_TRY(hipMemcpyAsync(h->d_counters, counters, sizeof(counters), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->d_done_count, 0, sizeof(int), s));
  HIP_TRY(hipMemsetAsync(h->d_finished, 0, sizeof(int) * batch, s));
  HIP_TRY(hipStreamSynchronize(s));
  int steps_run = 1;      // picks made: replayed decoder steps + the final pick
  if (max_new > 1) {
    const crispy_asr::TsKey key{h->dec_max_keys <= 128 ? 0 : h->dec_max_keys <= 256 ? 1 : 2, 2, batch, 1, 0, nullptr, 1};
    int ran = 0;
    rc = run_steps(h, key, batch, max_new - 1, [&]() -> int { return generation_body(h, batch, s); }, &ran);
    if (rc != CRISPY_OK) return rc;
    steps_run += ran;
  }
  // the last pick needs no further decoder step
  HIP_TRY(argmax_f32(h->d_logits, h->d_suppress, h->d_suppress_first, h->d_counters + 1, V, logits_ld(h), h->d_tok, h->d_tokens_all,
                     h->d_best, batch, s, h->eot, h->d_finished, h->d_done_count));
  std::vector<int> all((size_t)max_new * batch, h->eot);
  std::vector<float> best((size_t)max_new * batch, 0.f);
  HIP_TRY(hipMemcpyAsync(all.data(), h->d_tokens_all, (size_t)steps_run * batch * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(best.data(), h->d_best, (size_t)steps_run * batch * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int b = 0; b < batch; ++b) {
    int n = max_new;
    for (int i = 0; i < max_new; ++i) {
      tokens_out[(size_t)b * max_new + i] = all[(size_t)i * batch + b];
      if (logits_out) logits_out[(size_t)b * max_new + i] = best[(size_t)i * batch + b];
      if (n == max_new && all[(size_t)i * batch + b] == h->eot) n = i;
    }
    if (n_out) n_out[b] = n;
  }
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_decode_greedy_lang_device")

int crispy_asr_decode_timestamps_device(crispy_asr* h, const float* d_enc, int batch, const int* prompt, int n_prompt,
                                        const int* lang_tokens, int rules, const int* seek, const int* seek_end,
                                        int max_new, int* tokens_out, int* tids_out, int* n_out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_decode_timestamps_device: model not finalized");
  if (batch < 0 || max_new < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: negative size");
  if (batch == 0 || max_new == 0) return CRISPY_OK;
  if (!d_enc || !prompt || n_prompt <= 0 || !tokens_out)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: NULL argument");
  if (rules != TS_RULES_WCPP && rules != TS_RULES_OPENAI)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: rules must be 0 (whisper.cpp) or 1 (openai)");
  if (n_prompt + max_new > h->hp.n_text_ctx)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: %d prompt + %d new tokens exceed n_text_ctx %d",
                n_prompt, max_new, h->hp.n_text_ctx);
  if (special_tokens(h).beg + 1501 > h->hp.n_vocab)
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_decode_timestamps_device: vocabulary of %d has no timestamp tokens", h->hp.n_vocab);
  for (int i = 0; i < n_prompt; ++i)
    if (prompt[i] < 0 || prompt[i] >= h->hp.n_vocab)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_timestamps_device: prompt token %d out of range", prompt[i]);
  std::vector<std::vector<int>> prompts((size_t)batch, std::vector<int>(prompt, prompt + n_prompt));
  if (lang_tokens && n_prompt > 1)
    for (int b = 0; b < batch; ++b) prompts[b][1] = lang_tokens[b];
  return decode_ts(h, d_enc, batch, prompts, rules, seek, seek_end, max_new, h->d_suppress, h->d_suppress_first, 0.f, nullptr,
                   tokens_out, tids_out, nullptr, nullptr, n_out);
} CRISPY_CATCH_RET("crispy_asr_decode_timestamps_device")

int crispy_asr_decode_window_device(crispy_asr* h, const float* d_enc, int rows, const int* prompts, const int* n_prompt,
                                    int prompt_stride, int rules, const int* seek, const int* seek_end, int max_new,
                                    float temperature, const double* u, int* tokens_out, int* tids_out, float* plog_out,
                                    float* no_speech_prob_out, int* n_out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_decode_window_device: model not finalized");
  if (rows < 0 || max_new < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: negative size");
  if (rows == 0 || max_new == 0) return CRISPY_OK;
  if (!d_enc || !prompts || !n_prompt || !tokens_out || prompt_stride <= 0)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: NULL argument");
  if (rules != TS_RULES_WCPP && rules != TS_RULES_OPENAI)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: rules must be 0 (whisper.cpp) or 1 (openai)");
  if (special_tokens(h).beg + 1501 > h->hp.n_vocab)
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_decode_window_device: vocabulary of %d has no timestamp tokens", h->hp.n_vocab);
  if (u && !(temperature > 0.f))
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: sampling (u != NULL) needs a temperature > 0");
  std::vector<std::vector<int>> pr((size_t)rows);
  for (int b = 0; b < rows; ++b) {
    if (n_prompt[b] <= 0 || n_prompt[b] > prompt_stride)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_decode_window_device: row %d has a prompt of %d tokens (stride %d)", b, n_prompt[b], prompt_stride);
    pr[b].assign(prompts + (size_t)b * prompt_stride, prompts + (size_t)b * prompt_stride + n_prompt[b]);
  }
  if (!h->d_ts_mask) { const int rc = build_ts_masks(h); if (rc != CRISPY_OK) return rc; }
  return decode_ts(h, d_enc, rows, pr, rules, seek, seek_end, max_new, h->d_ts_mask, h->d_ts_mask_first, temperature, u,
                   tokens_out, tids_out, plog_out, no_speech_prob_out, n_out);
} CRISPY_CATCH_RET("crispy_asr_decode_window_device")

// whisper.cpp `whisper_lang_auto_detect`: feed <|startoftranscript|> alone and take the most probable
// language token [UPSTREAM-RECALL].  English-only vocabularies have nothing to detect.
int crispy_asr_detect_language_device(crispy_asr* h, const float* d_enc, int batch, int* lang_tokens_out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_detect_language_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_detect_language_device: model not finalized");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_detect_language_device: batch < 0");
  if (batch == 0) return CRISPY_OK;
  if (!d_enc || !lang_tokens_out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_detect_language_device: NULL argument");
  if (h->hp.n_vocab < 51865) return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_detect_language_device: English-only vocabulary");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  int rc = reserve_dec(h, batch);
  if (rc != CRISPY_OK) return rc;
  const int V = h->hp.n_vocab;
  rc = compute_cross_kv(h, d_enc, batch, s);
  if (rc != CRISPY_OK) return rc;
  const int sot = h->eot + 1, n_lang = 99 + (V - 51865);
  std::vector<int> tok(batch, sot);
  HIP_TRY(hipMemcpyAsync(h->d_tok, tok.data(), sizeof(int) * batch, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  h->dec_max_keys = 1;      // one position: the self K|V form (f16 in mode 1) must not depend on what the last decode call left here
  rc = decoder_step(h, batch, 0, false, true, s);
  if (rc != CRISPY_OK) return rc;
  if (!h->d_lang_mask) {
    std::vector<unsigned char> m(V, 1);
    for (int t = sot + 1; t < sot + 1 + n_lang && t < V; ++t) m[t] = 0;
    HIP_TRY(hipMalloc(&h->d_lang_mask, V));
    HIP_TRY(hipMemcpy(h->d_lang_mask, m.data(), V, hipMemcpyHostToDevice));
  }
  HIP_TRY(argmax_f32(h->d_logits, h->d_lang_mask, nullptr, nullptr, V, logits_ld(h), h->d_tok, nullptr, nullptr, batch, s));
  HIP_TRY(hipMemcpyAsync(lang_tokens_out, h->d_tok, sizeof(int) * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_detect_language_device")

int crispy_asr_transcribe_tokens(crispy_asr* h, const float* pcm, long pcm_stride, const int* n_samples, int batch,
                                 const int* prompt, int n_prompt, int max_new, int* tokens_out, int* n_out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_tokens: NULL handle");
  if (batch == 0) return CRISPY_OK;   // managers/transcription.rs:175-177: empty audio -> empty text
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_transcribe_tokens: model not finalized");
  if (!pcm || !n_samples || !tokens_out || !prompt)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_tokens: NULL argument");
  HIP_TRY(hipSetDevice(h->device));
  int rc = reserve_enc(h, batch);
  if (rc != CRISPY_OK) return rc;
  if (!h->w_pcm || pcm_stride > h->cap_pcm_stride) {
    if (h->w_pcm) (void)hipFree(h->w_pcm);
    h->w_pcm = nullptr;
    HIP_TRY(hipMalloc(&h->w_pcm, (size_t)h->cap_batch * pcm_stride * sizeof(float)));
    h->cap_pcm_stride = pcm_stride;
  }
  HIP_TRY(hipMemcpyAsync(h->w_pcm, pcm, (size_t)batch * pcm_stride * sizeof(float), hipMemcpyHostToDevice, h->stream));
  rc = crispy_mel_compute_device(h->mel, h->w_pcm, pcm_stride, n_samples, batch, nullptr, h->w_melt, h->stream);
  if (rc != CRISPY_OK) return rc;
  rc = crispy_asr_encode_device(h, h->w_melt, batch, h->w_enc, h->stream);
  if (rc != CRISPY_OK) return rc;
  return crispy_asr_decode_greedy_device(h, h->w_enc, batch, prompt, n_prompt, max_new, tokens_out, n_out, nullptr);
} CRISPY_CATCH_RET("crispy_asr_transcribe_tokens")


int crispy_asr_vocab_specials(int n_vocab, crispy_asr_specials* out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_vocab_specials: out is NULL");
  if (n_vocab < 51864) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_vocab_specials: %d is not a whisper vocabulary size", n_vocab);
  const Special sp = vocab_specials(n_vocab);
  out->eot = sp.sot - 1; out->sot = sp.sot; out->lang0 = sp.lang0; out->n_lang = sp.n_lang;
  out->translate = sp.translate; out->transcribe = sp.transcribe; out->solm = sp.solm; out->prev = sp.prev;
  out->nosp = sp.nosp; out->notimestamps = sp.not_; out->beg = sp.beg; out->multilingual = sp.multilingual ? 1 : 0;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_vocab_specials")

int crispy_asr_language_token(int n_vocab, const char* code, int* token_out) try {
  if (!code || !token_out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_language_token: NULL argument");
  if (n_vocab < 51864) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_language_token: %d is not a whisper vocabulary size", n_vocab);
  // whisper.cpp g_lang / openai LANGUAGES order [UPSTREAM-RECALL]; index = whisper_lang_id
  static const char* const kLang[] = {
      "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi",
      "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk",
      "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw",
      "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo",
      "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"};
  *token_out = 0;
  if (!*code || std::strcmp(code, "auto") == 0) return CRISPY_OK;
  const Special sp = vocab_specials(n_vocab);
  int id = -1;
  for (int i = 0; i < (int)(sizeof(kLang) / sizeof(kLang[0])); ++i)
    if (std::strcmp(code, kLang[i]) == 0) { id = i; break; }
  if (id < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_language_token: unknown language code '%s'", code);
  if (!sp.multilingual) {
    if (id == 0) return CRISPY_OK;                 // an English-only model transcribes English with no language token
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_language_token: an English-only vocabulary cannot take '%s'", code);
  }
  if (id >= sp.n_lang)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_language_token: this vocabulary has %d languages, '%s' is number %d", sp.n_lang, code, id + 1);
  *token_out = sp.lang0 + id;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_language_token")

int crispy_asr_token_text(const crispy_asr* h, int token, const char** text, size_t* len) try {
  if (!h || !text || !len) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_token_text: NULL argument");
  if (token < 0 || token >= (int)h->vocab.size())
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_token_text: token %d has no vocabulary entry", token);
  *text = h->vocab[token].data();
  *len = h->vocab[token].size();
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_token_text")

}  // extern "C"
