// whisper_api.cpp -- extern "C" Whisper entry points (include/crispy_hip.h): model container,
// encoder, greedy decoder.  Replaces transcribe_rs::whisper_cpp::WhisperEngine::{load, transcribe}
// (reference: src-tauri/src/managers/transcription.rs:138-141, 183-185).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "../../include/crispy_hip.h"
#include "api_util.h"
#include "asr_common.h"
#include "asr_quant.h"

using namespace crispy;

namespace {

struct Tensor {
  float* d = nullptr;
  size_t n = 0;
  bool set = false;
};

// A 2-D tensor kept in HBM as the model file holds it (ggml blocks, asr_quant.h; ttype QT_F32: a dense f32 tensor of a
// mixed file) and a row-wise concatenation of up to three of them (q | k | v, k | v): `crispy_asr_load_resident`.
struct QTensor {
  unsigned char* d = nullptr;
  int ttype = 0;
  size_t n = 0;          // elements
  int cols = 0;          // innermost dimension (K)
  size_t nbytes = 0;
  bool owned = true;     // false: d aliases a dense Tensor of the handle
};
struct QRef {
  const QTensor* t[3] = {nullptr, nullptr, nullptr};
  int n = 0;
  size_t elems() const { size_t e = 0; for (int i = 0; i < n; ++i) e += t[i]->n; return e; }
};

struct EncLayer {
  QRef r_qkv, r_out, r_fc1, r_fc2;           // resident model: the weights as quantised blocks
  const void *qkv_wh = nullptr, *out_wh = nullptr, *fc1_wh = nullptr, *fc2_wh = nullptr;   // f16 copies (precision mode 1)
  const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct DecLayer {
  QRef r_qkv, r_out, r_xq, r_xkv, r_xout, r_fc1, r_fc2;
  const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b;
  const float *lnx_w, *lnx_b, *xq_w, *xq_b, *xkv_w, *xkv_b, *xout_w, *xout_b;
  const void* xkv_wh = nullptr;              // f16 copy of the fused cross K|V projection (precision mode 1)
  const void *out_wh = nullptr, *xout_wh = nullptr, *fc2_wh = nullptr;   // f16 copies of the plain (no LayerNorm in front) decode projections
  const void *qkv_wh = nullptr, *xq_wh = nullptr, *fc1_wh = nullptr;     // f16 copies of the un-folded q | k | v, cross-q, fc1 (precision modes 1 / 2)
  const void *qkv_p = nullptr, *out_p = nullptr, *fc1_p = nullptr, *fc2_p = nullptr;   // ... packed for the fused step kernels (fused_pack_weights)
  const float *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  // LayerNorm folded into the consuming projection (decode steps with <= 64 clips): gamma-scaled weights, their row
  // sums and beta.W + bias (GemmArgs::ln_s / ln_c)
  const float *qkv_lw, *qkv_ls, *qkv_lc, *xq_lw, *xq_ls, *xq_lc, *fc1_lw, *fc1_ls, *fc1_lc;
};

}  // namespace

struct crispy_asr {
  int device = 0;
  crispy_asr_hparams hp{};
  hipStream_t stream = nullptr;
  crispy_mel* mel = nullptr;
  std::map<std::string, Tensor> tensors;   // as named by the model file
  std::vector<float*> derived;             // fused / reordered copies owned by the handle
  size_t derived_bytes = 0;                // ... and their size (crispy_asr_memory_info)
  // resident quantised model (crispy_asr_load_resident): 2-D tensors stay as ggml blocks, de-quantised into ONE scratch
  // slot right in front of the kernel that consumes them (same stream: the consumer has finished before the next fill)
  bool resident = false;
  std::map<std::string, QTensor> qtensors;
  void* q_scratch = nullptr;
  size_t q_scratch_bytes = 0;
  hipEvent_t ev_scratch = nullptr;           // orders a caller's stream against the handle's around the scratch slot
  const QTensor* q_tok_emb = nullptr;
  bool finalized = false;
  // resolved pointers
  const float *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr, *conv2_b = nullptr, *enc_pos = nullptr;
  const float *ln_post_w = nullptr, *ln_post_b = nullptr;
  const void* tok_emb_hp = nullptr;          // token embedding as f16 in MFMA operand order (precision mode 1: logits)
  const void* conv2_wh = nullptr;            // f16 copy of the reordered conv2 kernel (precision mode 1)
  const void* conv1_wh = nullptr;            // f16 conv1 kernel, rows zero-padded to conv1_kp columns
  int conv1_kp = 0;
  int enc_precision = 0;                     // 0: f32 operands (default), 1: f16 operands for the encoder GEMMs
  // precision modes 1 and 2: the decoder's LayerNorm output rounded to f16 in front of q | k | v, cross q and fc1 (f16 weights) --
  // ggml's mul_mat arithmetic for these products too [UPSTREAM-RECALL].  (Rounds 2 - 4 kept them in f32 in mode 1, with the
  // LayerNorm folded into an f32 GEMM; since round 5 a generated token runs through the fused step kernels of
  // whisper_dec_fused.hip, which multiply f16 LayerNorm outputs, and the staged path follows so that a position's
  // arithmetic does not depend on which path computed it.)
  bool dec_ln16 = false;
  bool dec_attn16 = false;                   // precision mode 2: + the query and the normalised probabilities rounded to f16 inside every attention
  bool ln16_ready = false;
  bool fused_path = true;                    // generated tokens through the fused step kernels when the model allows (CRISPY_ASR_DECODE=stages: never)
  float* d_fx[3] = {nullptr, nullptr, nullptr};      // fused step: residual stream after the self / cross / MLP input sums [rows][dt]
  float* d_fpart[3] = {nullptr, nullptr, nullptr};
  float* d_gvpart = nullptr;                 // gemv step (catalog widths): partial soft-maxes of the cross-attention [GEMV_MAX_M][heads][XA_PARTS][XA_PART_FLOATS]   // fused step: partial rows of the self / cross out-projection [heads][rows][dt], MLP [dt / 32][rows][dt]
  bool half_ready = false;                   // every f16 weight copy of mode 1 exists (set after the last one and a stream sync)
  int xcd_swizzle = 1;                       // mode 1 GEMMs: column tiles of a row tile on one XCD (CRISPY_ASR_XCD=0 turns it off)
  std::vector<EncLayer> enc;
  const float *tok_emb = nullptr, *dec_pos = nullptr, *dec_ln_w = nullptr, *dec_ln_b = nullptr;
  std::vector<DecLayer> dec;
  unsigned char* d_suppress = nullptr;      // [n_vocab] tokens never emitted by the greedy decoder
  unsigned char* d_suppress_first = nullptr;  // additionally suppressed at the first sampled position
  unsigned char* d_lang_mask = nullptr;       // everything but the language tokens (auto-detection)
  // workspace (grown on demand)
  int cap_batch = 0;
  float *w_melt = nullptr, *w_pcm = nullptr, *w_h1 = nullptr, *w_x = nullptr, *w_xn = nullptr, *w_qkv = nullptr,
        *w_att = nullptr, *w_h = nullptr, *w_enc = nullptr;
  long cap_pcm_stride = 0;
  // decoder workspace
  int dcap_batch = 0, dcap_xclips = 0;       // rows / audio clips the decoder workspace holds
  float *d_xkv = nullptr, *d_selfkv = nullptr, *d_dx = nullptr, *d_dxn = nullptr, *d_dq = nullptr, *d_datt = nullptr,
        *d_dh = nullptr, *d_logits = nullptr, *d_best = nullptr;
  int* d_tok = nullptr;
  int* d_tokens_all = nullptr;
  int* d_counters = nullptr;                 // [0] position, [1] generation step (device-side, advanced in-graph)
  // one captured decode step, replayed per generated token -- one per key class (<= 128 / 256 / 512 positions: the
  // self-attention kernel of mode 1 is baked into the capture).  A transcribe call with previous-text conditioning
  // alternates between classes from window to window (bare prompt, then prompt + past): with a single slot every window
  // re-instantiated the graph (1 - 2 ms each).

  int dec_max_keys = 0;                      // positions the current decode call can reach (prompt + new tokens)
  // timestamp-mode decoding (whisper.cpp no_timestamps = false)
  TsState* d_ts_state = nullptr;             // [dcap_batch]
  int* d_tids_all = nullptr;                 // [n_text_ctx][dcap_batch]
  int* d_done_count = nullptr;
  int* d_finished = nullptr;                 // [dcap_batch] plain greedy decoding: clip has produced its EOT
  void* d_xkv_h = nullptr;                   // f16 copy of the cross K|V (precision mode 1)
  unsigned char* d_ts_mask = nullptr;        // [n_vocab] whisper.cpp's always-suppressed specials
  unsigned char* d_ts_mask_first = nullptr;  // ... plus suppress_blank (" " and EOT) at the first position
  unsigned char* d_ts_mask_nst = nullptr;    // the two masks with whisper.cpp's non-speech tokens added (opts.suppress_nst; built on first use)
  unsigned char* d_ts_mask_first_nst = nullptr;
  std::vector<int> prompt_past;              // conditioning text the last single-chunk call ended with (opts.carry_context)
  // captured window-decode steps by what is baked into them: key class, kind of pick (greedy / sampling: different kernels),
  // rows, rows per clip, rules and mask.  A transcribe call alternates between several of them -- the greedy pass over all
  // clips, sampling passes over the failed ones x best_of, windows with and without the text so far -- and with one
  // slot per class every switch re-captured the step (1 - 2 ms each).
  struct TsKey {
    int kc, sampling, rows, xgroup, rules;     // sampling: 0 greedy pick under the timestamp rules, 1 sampling pick, 2 plain arg-max (no timestamps)
    const unsigned char* mask;
    int steps;                                 // generated tokens per replay
    bool operator<(const TsKey& o) const {
      if (steps != o.steps) return steps < o.steps;
      if (kc != o.kc) return kc < o.kc;
      if (sampling != o.sampling) return sampling < o.sampling;
      if (rows != o.rows) return rows < o.rows;
      if (xgroup != o.xgroup) return xgroup < o.xgroup;
      if (rules != o.rules) return rules < o.rules;
      return mask < o.mask;
    }
  };
  std::map<TsKey, hipGraphExec_t> ts_graphs;
  float* d_plog_all = nullptr;               // [n_text_ctx][dcap_batch] log-probability of every pick
  float* d_nosp = nullptr;                   // [dcap_batch] no_speech_prob of the window
  float* d_ts_x = nullptr;                   // [dcap_batch][TS_SCRATCH_ROW] the sampling pick's filtered rows
  double* d_u_all = nullptr;                 // [n_text_ctx][dcap_batch] uniform variates of a sampling pass (drawn on the host)
  float* d_temperature = nullptr;            // device scalar
  int* d_row_off = nullptr;                  // [dcap_batch] left padding of every clip's prompt (cache rows)
  const int* cur_row_off = nullptr;          // d_row_off while a window decode is running, else nullptr (decoder_step reads it)
  void* d_beam_kv = nullptr; size_t beam_kv_bytes = 0;      // beam search: the rows' cache bytes in flight between parents and children
  int* d_beam_parent = nullptr;              // [dcap_batch]
  int cur_xgroup = 1;                        // rows per audio clip while a window decode is running: the best-of decoders of a clip are
                                             // rows of their own (own self K|V cache) over ONE cross K|V (decode_ts)
  void drop_graphs() {
    for (auto& kv : ts_graphs)
      if (kv.second) (void)hipGraphExecDestroy(kv.second);
    ts_graphs.clear();
  }
  int eot = 50257;
  std::vector<unsigned char> sup_all, sup_first;   // host copies of the two suppression lists
  std::vector<std::string> vocab;                  // token byte strings of a loaded model file
};

namespace {

int build_ts_masks(crispy_asr* h);   // defined with the timestamp-mode code below
bool gemv_ref_ok(const QRef& r);     // defined with the decode steps below
int transcribe_batch_impl(crispy_asr* h, const float* const* pcm, const size_t* n, int batch, const crispy_asr_opts* opts,
                          crispy_asr_result** results, const volatile int* cancel);

void add_spec(std::map<std::string, size_t>& spec, const std::string& name, size_t n) { spec[name] = n; }

std::map<std::string, size_t> expected_tensors(const crispy_asr_hparams& hp) {
  std::map<std::string, size_t> s;
  const size_t d = hp.n_audio_state, dt = hp.n_text_state;
  add_spec(s, "encoder.conv1.weight", d * hp.n_mels * 3);
  add_spec(s, "encoder.conv1.bias", d);
  add_spec(s, "encoder.conv2.weight", d * d * 3);
  add_spec(s, "encoder.conv2.bias", d);
  add_spec(s, "encoder.positional_embedding", (size_t)hp.n_audio_ctx * d);
  for (int i = 0; i < hp.n_audio_layer; ++i) {
    const std::string p = "encoder.blocks." + std::to_string(i) + ".";
    add_spec(s, p + "attn_ln.weight", d); add_spec(s, p + "attn_ln.bias", d);
    add_spec(s, p + "attn.query.weight", d * d); add_spec(s, p + "attn.query.bias", d);
    add_spec(s, p + "attn.key.weight", d * d);
    add_spec(s, p + "attn.value.weight", d * d); add_spec(s, p + "attn.value.bias", d);
    add_spec(s, p + "attn.out.weight", d * d); add_spec(s, p + "attn.out.bias", d);
    add_spec(s, p + "mlp_ln.weight", d); add_spec(s, p + "mlp_ln.bias", d);
    add_spec(s, p + "mlp.0.weight", 4 * d * d); add_spec(s, p + "mlp.0.bias", 4 * d);
    add_spec(s, p + "mlp.2.weight", 4 * d * d); add_spec(s, p + "mlp.2.bias", d);
  }
  add_spec(s, "encoder.ln_post.weight", d); add_spec(s, "encoder.ln_post.bias", d);
  add_spec(s, "decoder.token_embedding.weight", (size_t)hp.n_vocab * dt);
  add_spec(s, "decoder.positional_embedding", (size_t)hp.n_text_ctx * dt);
  for (int i = 0; i < hp.n_text_layer; ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i) + ".";
    for (const char* a : {"attn", "cross_attn"}) {
      const std::string q = p + a;
      add_spec(s, q + "_ln.weight", dt); add_spec(s, q + "_ln.bias", dt);
      add_spec(s, q + ".query.weight", dt * dt); add_spec(s, q + ".query.bias", dt);
      add_spec(s, q + ".key.weight", dt * dt);
      add_spec(s, q + ".value.weight", dt * dt); add_spec(s, q + ".value.bias", dt);
      add_spec(s, q + ".out.weight", dt * dt); add_spec(s, q + ".out.bias", dt);
    }
    add_spec(s, p + "mlp_ln.weight", dt); add_spec(s, p + "mlp_ln.bias", dt);
    add_spec(s, p + "mlp.0.weight", 4 * dt * dt); add_spec(s, p + "mlp.0.bias", 4 * dt);
    add_spec(s, p + "mlp.2.weight", 4 * dt * dt); add_spec(s, p + "mlp.2.bias", dt);
  }
  add_spec(s, "decoder.ln.weight", dt); add_spec(s, "decoder.ln.bias", dt);
  return s;
}

int upload(crispy_asr* h, const std::vector<float>& host, const float** out) {
  float* d = nullptr;
  HIP_TRY(hipMalloc(&d, host.size() * sizeof(float)));
  h->derived.push_back(d);
  h->derived_bytes += host.size() * sizeof(float);
  HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = d;
  return CRISPY_OK;
}

int download(const Tensor& t, std::vector<float>& host) {
  host.resize(t.n);
  HIP_TRY(hipMemcpy(host.data(), t.d, t.n * sizeof(float), hipMemcpyDeviceToHost));
  return CRISPY_OK;
}

const float* T(crispy_asr* h, const std::string& name) { return h->tensors[name].d; }

// [co][ci][3] -> [co][kk*ci_n + ci]: K order (tap, channel) matches three consecutive frame-major rows
int reorder_conv(crispy_asr* h, const std::string& name, int co_n, int ci_n, const float** out) {
  std::vector<float> src, dst((size_t)co_n * ci_n * 3);
  int rc = download(h->tensors[name], src);
  if (rc != CRISPY_OK) return rc;
  for (int co = 0; co < co_n; ++co)
    for (int ci = 0; ci < ci_n; ++ci)
      for (int kk = 0; kk < 3; ++kk) dst[((size_t)co * 3 + kk) * ci_n + ci] = src[((size_t)co * ci_n + ci) * 3 + kk];
  return upload(h, dst, out);
}

// concatenate row blocks of [d][d] weights (and biases; a missing bias is zeros)
int fuse_rows(crispy_asr* h, const std::vector<std::string>& wnames, const std::vector<std::string>& bnames, int d,
              const float** w_out, const float** b_out) {
  std::vector<float> W, Bv;
  for (size_t i = 0; i < wnames.size(); ++i) {
    std::vector<float> t;
    int rc = download(h->tensors[wnames[i]], t);
    if (rc != CRISPY_OK) return rc;
    W.insert(W.end(), t.begin(), t.end());
    if (!bnames[i].empty()) {
      rc = download(h->tensors[bnames[i]], t);
      if (rc != CRISPY_OK) return rc;
      Bv.insert(Bv.end(), t.begin(), t.end());
    } else {
      Bv.insert(Bv.end(), (size_t)d, 0.f);
    }
  }
  int rc = upload(h, W, w_out);
  if (rc != CRISPY_OK) return rc;
  return upload(h, Bv, b_out);
}

// LayerNorm (gamma, beta over K) folded into W [N][K] (+ bias [N], may be null): see GemmArgs::ln_s
int fold_ln(crispy_asr* h, const float* d_w, const float* d_bias, const float* d_gamma, const float* d_beta, size_t N,
            size_t K, const float** lw, const float** ls, const float** lc) {
  std::vector<float> W(N * K), b(N, 0.f), g(K), be(K);
  HIP_TRY(hipMemcpy(W.data(), d_w, W.size() * sizeof(float), hipMemcpyDeviceToHost));
  if (d_bias) HIP_TRY(hipMemcpy(b.data(), d_bias, N * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(g.data(), d_gamma, K * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(be.data(), d_beta, K * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<float> s(N), c(N);
  for (size_t n = 0; n < N; ++n) {
    double ss = 0.0, cc = (double)b[n];
    float* row = W.data() + n * K;
    for (size_t k = 0; k < K; ++k) {
      cc += (double)be[k] * (double)row[k];
      row[k] *= g[k];
      ss += (double)row[k];
    }
    s[n] = (float)ss;
    c[n] = (float)cc;
  }
  int rc = upload(h, W, lw);
  if (rc == CRISPY_OK) rc = upload(h, s, ls);
  if (rc == CRISPY_OK) rc = upload(h, c, lc);
  return rc;
}

// ---- resident quantised tensors (asr_quant.h) -----------------------------------------------------------------
// Dense copy of a (row-concatenated) resident tensor in the handle's scratch slot, enqueued on `s` right in front of
// its consumer: f16 (the operands of precision mode 1), f32, or f32 x gamma[k] (the LayerNorm-folded decode projections).
int dq(crispy_asr* h, const QRef& r, bool f16, const float* gamma, hipStream_t s, const void** out) {
  const size_t esz = f16 ? 2 : 4;
  if (r.n <= 0 || r.elems() * esz > h->q_scratch_bytes)
    return fail(CRISPY_ERR_INVALID_ARG, "resident model: tensor of %zu elements does not fit the de-quantisation slot", r.elems());
  char* dst = reinterpret_cast<char*>(h->q_scratch);
  for (int i = 0; i < r.n; ++i) {
    const QTensor& t = *r.t[i];
    HIP_TRY(dequant_blocks(t.d, t.ttype, (long)(t.n / 32), t.cols, dst, f16 ? 1 : 0, gamma, s));
    dst += t.n * esz;
  }
  *out = h->q_scratch;
  return CRISPY_OK;
}

QRef qref(crispy_asr* h, std::initializer_list<std::string> names) {
  QRef r;
  for (const std::string& n : names) r.t[r.n++] = &h->qtensors[n];
  return r;
}

// concatenated biases of row-fused weights (a missing bias is zeros)
int fuse_bias(crispy_asr* h, const std::vector<std::string>& bnames, int d, const float** b_out) {
  std::vector<float> Bv;
  for (const std::string& b : bnames) {
    if (!b.empty()) {
      std::vector<float> t;
      const int rc = download(h->tensors[b], t);
      if (rc != CRISPY_OK) return rc;
      Bv.insert(Bv.end(), t.begin(), t.end());
    } else {
      Bv.insert(Bv.end(), (size_t)d, 0.f);
    }
  }
  return upload(h, Bv, b_out);
}

void free_ws(crispy_asr* h) {
  for (float** p : {&h->w_melt, &h->w_pcm, &h->w_h1, &h->w_x, &h->w_xn, &h->w_qkv, &h->w_att, &h->w_h, &h->w_enc})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  h->cap_batch = 0;
  h->cap_pcm_stride = 0;
}
void free_dec_ws(crispy_asr* h) {
  for (float** p : {&h->d_xkv, &h->d_selfkv, &h->d_dx, &h->d_dxn, &h->d_dq, &h->d_datt, &h->d_dh, &h->d_logits, &h->d_best})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->d_tok) { (void)hipFree(h->d_tok); h->d_tok = nullptr; }
  if (h->d_tokens_all) { (void)hipFree(h->d_tokens_all); h->d_tokens_all = nullptr; }
  if (h->d_counters) { (void)hipFree(h->d_counters); h->d_counters = nullptr; }
  h->drop_graphs();
  if (h->d_ts_state) { (void)hipFree(h->d_ts_state); h->d_ts_state = nullptr; }
  if (h->d_tids_all) { (void)hipFree(h->d_tids_all); h->d_tids_all = nullptr; }
  if (h->d_done_count) { (void)hipFree(h->d_done_count); h->d_done_count = nullptr; }
  if (h->d_finished) { (void)hipFree(h->d_finished); h->d_finished = nullptr; }
  if (h->d_xkv_h) { (void)hipFree(h->d_xkv_h); h->d_xkv_h = nullptr; }
  if (h->d_plog_all) { (void)hipFree(h->d_plog_all); h->d_plog_all = nullptr; }
  if (h->d_nosp) { (void)hipFree(h->d_nosp); h->d_nosp = nullptr; }
  if (h->d_u_all) { (void)hipFree(h->d_u_all); h->d_u_all = nullptr; }
  if (h->d_ts_x) { (void)hipFree(h->d_ts_x); h->d_ts_x = nullptr; }
  if (h->d_beam_kv) { (void)hipFree(h->d_beam_kv); h->d_beam_kv = nullptr; h->beam_kv_bytes = 0; }
  if (h->d_beam_parent) { (void)hipFree(h->d_beam_parent); h->d_beam_parent = nullptr; }
  if (h->d_temperature) { (void)hipFree(h->d_temperature); h->d_temperature = nullptr; }
  if (h->d_row_off) { (void)hipFree(h->d_row_off); h->d_row_off = nullptr; }
  if (h->d_gvpart) { (void)hipFree(h->d_gvpart); h->d_gvpart = nullptr; }
  for (int i = 0; i < 3; ++i) {
    if (h->d_fx[i]) { (void)hipFree(h->d_fx[i]); h->d_fx[i] = nullptr; }
    if (h->d_fpart[i]) { (void)hipFree(h->d_fpart[i]); h->d_fpart[i] = nullptr; }
  }
  h->cur_row_off = nullptr;
  h->dcap_batch = 0;
  h->dcap_xclips = 0;
}

int reserve_enc(crispy_asr* h, int batch) {
  if (batch <= h->cap_batch) return CRISPY_OK;
  free_ws(h);
  const size_t B = batch, d = h->hp.n_audio_state, Tn = h->hp.n_audio_ctx;
  HIP_TRY(hipMalloc(&h->w_melt, B * (MEL_FRAMES + 2) * h->hp.n_mels * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_h1, B * (MEL_FRAMES + 1) * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_x, B * Tn * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_xn, B * Tn * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_qkv, B * Tn * 3 * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_att, B * Tn * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_h, B * Tn * 4 * d * sizeof(float)));
  HIP_TRY(hipMalloc(&h->w_enc, B * Tn * d * sizeof(float)));
  // zero padding rows of the frame-major buffers are written once
  HIP_TRY(hipMemset(h->w_melt, 0, B * (MEL_FRAMES + 2) * h->hp.n_mels * sizeof(float)));
  HIP_TRY(hipMemset(h->w_h1, 0, B * (MEL_FRAMES + 1) * d * sizeof(float)));
  // hipMemset on device memory does not wait on the host, and it runs on the NULL stream, which the handle's
  // non-blocking stream is not ordered against: without this the 1.5 GB memset of a 256-clip workspace was still
  // clearing h1 while the first call's conv1 had already written the first clips (wrong encoder output for clips
  // 0..7 of the first 256-clip call, intermittently: tests/test_gpu_pipeline.py cfg5)
  HIP_TRY(hipDeviceSynchronize());
  h->cap_batch = batch;
  return CRISPY_OK;
}

GemmArgs gemm(const float* A, long lda, const float* W, long ldw, float* C, long ldc, const float* bias, int M, int N,
              int K) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  return g;
}

}  // namespace

extern "C" {

int crispy_asr_create(const crispy_asr_hparams* hp, const float* mel_filters, int device, crispy_asr** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_create: out is NULL");
  *out = nullptr;
  if (!hp || !mel_filters) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_create: NULL argument");
  auto width_ok = [](int d) { return d == 384 || d == 512 || d == 768 || d == 1024 || d == 1280; };   // tiny ... large
  if (hp->n_audio_ctx != 1500 || !width_ok(hp->n_audio_state) || !width_ok(hp->n_text_state) ||
      // (divisions, not products: a hostile head count from a model file must not overflow -- found by the sanitizer
      // harness, tests/test_host_sanitizers.py)
      hp->n_audio_state % 64 != 0 || hp->n_audio_head != hp->n_audio_state / 64 || hp->n_text_head != hp->n_text_state / 64 ||
      hp->n_audio_layer <= 0 || hp->n_audio_layer > 64 || hp->n_text_layer <= 0 || hp->n_text_layer > 64 ||
      hp->n_text_ctx <= 0 || hp->n_text_ctx > 448 || hp->n_vocab <= 0 || (hp->n_mels != 80 && hp->n_mels != 128) ||
      (hp->n_mels * 3) % 16)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_create: unsupported hyper-parameters (widths 384/512/768/1024/1280, head dim 64, ctx 1500/<=448)");
  int rc = check_device(device, "crispy_asr_create");
  if (rc != CRISPY_OK) return rc;
  crispy_asr* h = new (std::nothrow) crispy_asr();
  if (!h) return fail(CRISPY_ERR_OOM, "crispy_asr_create: host allocation failed");
  h->device = device;
  h->hp = *hp;
  h->eot = hp->n_vocab >= 51865 ? 50257 : 50256;   // multilingual vocabularies shift the specials by one
  if (const char* e = dev_env("CRISPY_ASR_XCD")) h->xcd_swizzle = std::atoi(e) != 0;
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    int r = crispy_mel_create(mel_filters, hp->n_mels, device, &h->mel);
    if (r != CRISPY_OK) return r;
    for (const auto& kv : expected_tensors(*hp)) {     // device memory is taken when a tensor is set: a resident
      Tensor t;                                         // quantised model never holds its matrices as f32
      t.n = kv.second;
      h->tensors[kv.first] = t;
    }
    return CRISPY_OK;
  };
  rc = body();
  if (rc != CRISPY_OK) {
    const std::string keep = last_error_cstr();
    crispy_asr_free(h);
    return fail(rc, "%s", keep.c_str());
  }
  *out = h;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_create")

void crispy_asr_free(crispy_asr* h) try {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& kv : h->tensors)
    if (kv.second.d) (void)hipFree(kv.second.d);
  for (float* p : h->derived) (void)hipFree(p);
  for (auto& kv : h->qtensors)
    if (kv.second.d && kv.second.owned) (void)hipFree(kv.second.d);
  if (h->q_scratch) (void)hipFree(h->q_scratch);
  if (h->ev_scratch) (void)hipEventDestroy(h->ev_scratch);
  if (h->d_suppress) (void)hipFree(h->d_suppress);
  if (h->d_suppress_first) (void)hipFree(h->d_suppress_first);
  if (h->d_ts_mask) (void)hipFree(h->d_ts_mask);
  if (h->d_ts_mask_first) (void)hipFree(h->d_ts_mask_first);
  if (h->d_ts_mask_nst) (void)hipFree(h->d_ts_mask_nst);
  if (h->d_ts_mask_first_nst) (void)hipFree(h->d_ts_mask_first_nst);
  if (h->d_lang_mask) (void)hipFree(h->d_lang_mask);
  free_ws(h);
  free_dec_ws(h);
  if (h->mel) crispy_mel_destroy(h->mel);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
} CRISPY_CATCH_VOID("crispy_asr_free")

int crispy_asr_set_tensor(crispy_asr* h, const char* name, const float* data, size_t n_elems) try {
  if (!h || !name || !data) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_tensor: NULL argument");
  if (h->finalized) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_tensor: model already finalized");
  auto it = h->tensors.find(name);
  if (it == h->tensors.end()) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_set_tensor: unknown tensor '%s'", name);
  if (it->second.n != n_elems)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_set_tensor: '%s' has %zu elements, expected %zu", name, n_elems,
                it->second.n);
  HIP_TRY(hipSetDevice(h->device));
  if (!it->second.d) HIP_TRY(hipMalloc(&it->second.d, n_elems * sizeof(float)));
  HIP_TRY(hipMemcpy(it->second.d, data, n_elems * sizeof(float), hipMemcpyHostToDevice));
  it->second.set = true;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_set_tensor")

namespace {

int finalize_tail(crispy_asr* h) {
  HIP_TRY(hipMalloc(&h->d_suppress, h->hp.n_vocab));
  HIP_TRY(hipMalloc(&h->d_suppress_first, h->hp.n_vocab));
  HIP_TRY(hipMemset(h->d_suppress, 0, h->hp.n_vocab));
  HIP_TRY(hipMemset(h->d_suppress_first, 0, h->hp.n_vocab));
  HIP_TRY(hipDeviceSynchronize());          // NULL-stream memsets vs the handle's non-blocking stream (see reserve_enc)
  { const int mrc = build_ts_masks(h); if (mrc != CRISPY_OK) return mrc; }
  h->finalized = true;
  return CRISPY_OK;
}

// f16 copies of the two convolution kernels (precision mode 1; conv1's rows zero-padded to a multiple of 32 columns: the
// padded operand columns of A read on into the next frames -- finite values -- and meet zeros here)
int make_conv_halves(crispy_asr* h) {
  const size_t d = h->hp.n_audio_state;
  {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, d * 3 * d * 2));
    h->derived.push_back(reinterpret_cast<float*>(p));
    h->derived_bytes += d * 3 * d * 2;
    HIP_TRY(convert_f32_to_f16(h->conv2_w, p, (long)(d * 3 * d), h->stream));
    h->conv2_wh = p;
  }
  const int k1 = 3 * h->hp.n_mels, k1p = (k1 + 31) / 32 * 32;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, d * k1p * 2));
  h->derived.push_back(reinterpret_cast<float*>(p));
  h->derived_bytes += d * k1p * 2;
  HIP_TRY(hipMemsetAsync(p, 0, d * k1p * 2, h->stream));
  HIP_TRY(convert_rows_f32_to_f16(h->conv1_w, k1, p, k1p, k1, (long)d, h->stream));
  h->conv1_wh = p;
  h->conv1_kp = k1p;
  return CRISPY_OK;
}

// crispy_asr_load_resident: the 2-D tensors stay as the file's ggml blocks (h->qtensors); what is made here is small --
// fused biases, the LayerNorm-fold vectors of the decode projections, the two convolution kernels (f16 / f32 in every
// ggml file) -- plus the token embedding packed in MFMA operand order for the logits (f16: the one matrix kept dense;
// its quantised form stays resident too and serves the embedding look-ups).  Precision mode 1 only: the operands of
// every matrix product are de-quantised to f16 (f32 x gamma for the folded projections) right in front of the product.
int finalize_resident(crispy_asr* h) {
  const int d = h->hp.n_audio_state, dt = h->hp.n_text_state, V = h->hp.n_vocab;
  if (!(dt == 384 || dt == 512 || dt == 768 || dt == 1024 || dt == 1280))
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_load_resident: width %d has no f16 logits kernel", dt);
  int rc;
  if ((rc = reorder_conv(h, "encoder.conv1.weight", d, h->hp.n_mels, &h->conv1_w)) != CRISPY_OK) return rc;
  if ((rc = reorder_conv(h, "encoder.conv2.weight", d, d, &h->conv2_w)) != CRISPY_OK) return rc;
  h->conv1_b = T(h, "encoder.conv1.bias");
  h->conv2_b = T(h, "encoder.conv2.bias");
  h->enc_pos = T(h, "encoder.positional_embedding");
  h->ln_post_w = T(h, "encoder.ln_post.weight");
  h->ln_post_b = T(h, "encoder.ln_post.bias");
  // a matrix the file holds dense (f32 / f16: mixed files) takes part as a QT_F32 "block" tensor aliasing its dense copy
  auto matrix = [&](const std::string& name, int cols) -> int {
    if (h->qtensors.count(name)) return CRISPY_OK;
    Tensor& t = h->tensors[name];
    if (!t.d) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load_resident: tensor '%s' missing", name.c_str());
    QTensor q;
    q.d = reinterpret_cast<unsigned char*>(t.d); q.ttype = QT_F32; q.n = t.n; q.cols = cols; q.nbytes = t.n * 4; q.owned = false;
    h->qtensors[name] = q;
    return CRISPY_OK;
  };
  size_t max_elems = 0;
  auto ref = [&](std::initializer_list<std::string> names, int cols, QRef* out) -> int {
    for (const std::string& n : names) { const int r = matrix(n, cols); if (r != CRISPY_OK) return r; }
    *out = qref(h, names);
    if (out->elems() > max_elems) max_elems = out->elems();
    return CRISPY_OK;
  };
  h->enc.resize(h->hp.n_audio_layer);
  h->dec.resize(h->hp.n_text_layer);
  for (int i = 0; i < h->hp.n_audio_layer; ++i) {
    const std::string p = "encoder.blocks." + std::to_string(i) + ".";
    EncLayer& L = h->enc[i];
    L.ln1_w = T(h, p + "attn_ln.weight"); L.ln1_b = T(h, p + "attn_ln.bias");
    L.ln2_w = T(h, p + "mlp_ln.weight"); L.ln2_b = T(h, p + "mlp_ln.bias");
    L.out_b = T(h, p + "attn.out.bias"); L.fc1_b = T(h, p + "mlp.0.bias"); L.fc2_b = T(h, p + "mlp.2.bias");
    L.qkv_w = L.out_w = L.fc1_w = L.fc2_w = nullptr;
    if ((rc = ref({p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"}, d, &L.r_qkv)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "attn.out.weight"}, d, &L.r_out)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "mlp.0.weight"}, d, &L.r_fc1)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "mlp.2.weight"}, 4 * d, &L.r_fc2)) != CRISPY_OK) return rc;
    if ((rc = fuse_bias(h, {p + "attn.query.bias", "", p + "attn.value.bias"}, d, &L.qkv_b)) != CRISPY_OK) return rc;
  }
  for (int i = 0; i < h->hp.n_text_layer; ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i) + ".";
    DecLayer& L = h->dec[i];
    L.ln1_w = T(h, p + "attn_ln.weight"); L.ln1_b = T(h, p + "attn_ln.bias");
    L.lnx_w = T(h, p + "cross_attn_ln.weight"); L.lnx_b = T(h, p + "cross_attn_ln.bias");
    L.ln2_w = T(h, p + "mlp_ln.weight"); L.ln2_b = T(h, p + "mlp_ln.bias");
    L.out_b = T(h, p + "attn.out.bias"); L.xq_b = T(h, p + "cross_attn.query.bias");
    L.xout_b = T(h, p + "cross_attn.out.bias"); L.fc1_b = T(h, p + "mlp.0.bias"); L.fc2_b = T(h, p + "mlp.2.bias");
    L.qkv_w = L.out_w = L.xq_w = L.xkv_w = L.xout_w = L.fc1_w = L.fc2_w = nullptr;
    L.qkv_lw = L.xq_lw = L.fc1_lw = nullptr;
    if ((rc = ref({p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"}, dt, &L.r_qkv)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "attn.out.weight"}, dt, &L.r_out)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "cross_attn.query.weight"}, dt, &L.r_xq)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "cross_attn.key.weight", p + "cross_attn.value.weight"}, dt, &L.r_xkv)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "cross_attn.out.weight"}, dt, &L.r_xout)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "mlp.0.weight"}, dt, &L.r_fc1)) != CRISPY_OK) return rc;
    if ((rc = ref({p + "mlp.2.weight"}, 4 * dt, &L.r_fc2)) != CRISPY_OK) return rc;
    if ((rc = fuse_bias(h, {p + "attn.query.bias", "", p + "attn.value.bias"}, dt, &L.qkv_b)) != CRISPY_OK) return rc;
    if ((rc = fuse_bias(h, {"", p + "cross_attn.value.bias"}, dt, &L.xkv_b)) != CRISPY_OK) return rc;
  }
  if ((rc = matrix("decoder.token_embedding.weight", dt)) != CRISPY_OK) return rc;
  h->q_tok_emb = &h->qtensors["decoder.token_embedding.weight"];
  h->tok_emb = nullptr;
  h->dec_pos = T(h, "decoder.positional_embedding");
  h->dec_ln_w = T(h, "decoder.ln.weight");
  h->dec_ln_b = T(h, "decoder.ln.bias");
  // the one scratch slot: the largest (fused) matrix as f32
  h->q_scratch_bytes = max_elems * sizeof(float);
  HIP_TRY(hipMalloc(&h->q_scratch, h->q_scratch_bytes));
  {   // token embedding for the logits: f16, packed in MFMA operand order, from a dense copy that lives for this block only
    float* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)V * dt * sizeof(float)));
    hipError_t e = dequant_blocks(h->q_tok_emb->d, h->q_tok_emb->ttype, (long)((size_t)V * dt / 32), dt, tmp, 0, nullptr, h->stream);
    void* packed = nullptr;
    if (e == hipSuccess) e = hipMalloc(&packed, vocab_f16_packed_bytes(V, dt));
    if (e == hipSuccess) {
      h->derived.push_back(reinterpret_cast<float*>(packed));
      h->derived_bytes += vocab_f16_packed_bytes(V, dt);
      e = pack_vocab_f16(tmp, packed, V, dt, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    HIP_TRY(e);
    h->tok_emb_hp = packed;
  }
  if ((rc = make_conv_halves(h)) != CRISPY_OK) return rc;
  // One matrix per decoder layer is kept de-quantised as well: the cross-q projection, as f16 (the values the blocks
  // de-quantise to at the point of use: same bits).  The matrix-vector decode step computes a head's query in every one of the
  // workgroups that share the head's keys (whisper_dec_gemv.hip: gv_xattn_kernel), and de-quantising it four times over cost
  // large-v3-q5_0 more than the launch the fusion saves (1.78 -> 1.92 ms per token); 1 / 14 of the decoder's weights, + 7 % memory.
  if (gemv_dec_supported(dt, 1)) {
    for (DecLayer& L : h->dec) {
      if (!gemv_ref_ok(L.r_xq)) continue;
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, (size_t)dt * dt * 2));
      h->derived.push_back(reinterpret_cast<float*>(p));
      h->derived_bytes += (size_t)dt * dt * 2;
      HIP_TRY(dequant_blocks(L.r_xq.t[0]->d, L.r_xq.t[0]->ttype, (long)((size_t)dt * dt / 32), dt, p, 1, nullptr, h->stream));
      L.xq_wh = p;
    }
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->half_ready = true;
  h->enc_precision = 1;
  h->dec_ln16 = true;                        // mode 1: f16 LayerNorm outputs against the blocks de-quantised to f16 (gemm_skinny_q, WH forms)
  return finalize_tail(h);
}

}  // namespace

int crispy_asr_finalize(crispy_asr* h) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_finalize: NULL handle");
  if (h->finalized) return CRISPY_OK;
  for (const auto& kv : h->tensors)
    if (!kv.second.set) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_finalize: tensor '%s' was never set", kv.first.c_str());
  HIP_TRY(hipSetDevice(h->device));
  if (h->resident) return finalize_resident(h);
  const int d = h->hp.n_audio_state, dt = h->hp.n_text_state;
  int rc;
  if ((rc = reorder_conv(h, "encoder.conv1.weight", d, h->hp.n_mels, &h->conv1_w)) != CRISPY_OK) return rc;
  if ((rc = reorder_conv(h, "encoder.conv2.weight", d, d, &h->conv2_w)) != CRISPY_OK) return rc;
  h->conv1_b = T(h, "encoder.conv1.bias");
  h->conv2_b = T(h, "encoder.conv2.bias");
  h->enc_pos = T(h, "encoder.positional_embedding");
  h->ln_post_w = T(h, "encoder.ln_post.weight");
  h->ln_post_b = T(h, "encoder.ln_post.bias");
  h->enc.resize(h->hp.n_audio_layer);
  for (int i = 0; i < h->hp.n_audio_layer; ++i) {
    const std::string p = "encoder.blocks." + std::to_string(i) + ".";
    EncLayer& L = h->enc[i];
    L.ln1_w = T(h, p + "attn_ln.weight"); L.ln1_b = T(h, p + "attn_ln.bias");
    rc = fuse_rows(h, {p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"},
                   {p + "attn.query.bias", "", p + "attn.value.bias"}, d, &L.qkv_w, &L.qkv_b);
    if (rc != CRISPY_OK) return rc;
    L.out_w = T(h, p + "attn.out.weight"); L.out_b = T(h, p + "attn.out.bias");
    L.ln2_w = T(h, p + "mlp_ln.weight"); L.ln2_b = T(h, p + "mlp_ln.bias");
    L.fc1_w = T(h, p + "mlp.0.weight"); L.fc1_b = T(h, p + "mlp.0.bias");
    L.fc2_w = T(h, p + "mlp.2.weight"); L.fc2_b = T(h, p + "mlp.2.bias");
  }
  h->tok_emb = T(h, "decoder.token_embedding.weight");
  h->dec_pos = T(h, "decoder.positional_embedding");
  h->dec_ln_w = T(h, "decoder.ln.weight");
  h->dec_ln_b = T(h, "decoder.ln.bias");
  h->dec.resize(h->hp.n_text_layer);
  for (int i = 0; i < h->hp.n_text_layer; ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i) + ".";
    DecLayer& L = h->dec[i];
    L.ln1_w = T(h, p + "attn_ln.weight"); L.ln1_b = T(h, p + "attn_ln.bias");
    rc = fuse_rows(h, {p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight"},
                   {p + "attn.query.bias", "", p + "attn.value.bias"}, dt, &L.qkv_w, &L.qkv_b);
    if (rc != CRISPY_OK) return rc;
    L.out_w = T(h, p + "attn.out.weight"); L.out_b = T(h, p + "attn.out.bias");
    L.lnx_w = T(h, p + "cross_attn_ln.weight"); L.lnx_b = T(h, p + "cross_attn_ln.bias");
    L.xq_w = T(h, p + "cross_attn.query.weight"); L.xq_b = T(h, p + "cross_attn.query.bias");
    rc = fuse_rows(h, {p + "cross_attn.key.weight", p + "cross_attn.value.weight"}, {"", p + "cross_attn.value.bias"},
                   dt, &L.xkv_w, &L.xkv_b);
    if (rc != CRISPY_OK) return rc;
    L.xout_w = T(h, p + "cross_attn.out.weight"); L.xout_b = T(h, p + "cross_attn.out.bias");
    L.ln2_w = T(h, p + "mlp_ln.weight"); L.ln2_b = T(h, p + "mlp_ln.bias");
    L.fc1_w = T(h, p + "mlp.0.weight"); L.fc1_b = T(h, p + "mlp.0.bias");
    L.fc2_w = T(h, p + "mlp.2.weight"); L.fc2_b = T(h, p + "mlp.2.bias");
    rc = fold_ln(h, L.qkv_w, L.qkv_b, L.ln1_w, L.ln1_b, 3 * (size_t)dt, dt, &L.qkv_lw, &L.qkv_ls, &L.qkv_lc);
    if (rc == CRISPY_OK) rc = fold_ln(h, L.xq_w, L.xq_b, L.lnx_w, L.lnx_b, dt, dt, &L.xq_lw, &L.xq_ls, &L.xq_lc);
    if (rc == CRISPY_OK) rc = fold_ln(h, L.fc1_w, L.fc1_b, L.ln2_w, L.ln2_b, 4 * (size_t)dt, dt, &L.fc1_lw, &L.fc1_ls, &L.fc1_lc);
    if (rc != CRISPY_OK) return rc;
  }
  return finalize_tail(h);
} CRISPY_CATCH_RET("crispy_asr_finalize")

int crispy_asr_hparams_get(const crispy_asr* h, crispy_asr_hparams* out) try {
  if (!h || !out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_hparams_get: NULL argument");
  *out = h->hp;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_hparams_get")

// mel (frame-major, padded) -> encoder output [B][1500][d]
int crispy_asr_set_precision(crispy_asr* h, int mode) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_precision: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_set_precision: model not finalized");
  if (mode < 0 || mode > 2)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_set_precision: mode must be 0 (f32 operands), 1 (whisper.cpp's arithmetic: f16 operands of every product, f16 LayerNorm outputs and caches) "
                "or 2 (1 + the query and the normalised probabilities rounded to f16 inside the attentions)");
  HIP_TRY(hipSetDevice(h->device));
  const bool want_attn16 = mode == 2;
  if (mode == 2) mode = 1;
  const bool want_ln16 = mode == 1;
  if (h->resident && (mode != 1 || want_attn16))
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_set_precision: a resident quantised model runs in precision mode 1 only "
                "(its matrices exist as f16 operands at the point of use, never as f32 tensors)");
  if (mode == 1 && !h->half_ready) {
    // f16 copies of the encoder GEMM weights, made on the device once.  `half_ready` is only set after the last copy
    // and a stream sync: a hipMalloc failing part-way (OOM on a large model) leaves the mode at 0 and a retry starts
    // over (the partial copies stay owned by `derived` until the handle is freed) -- ADVICE r2.
    const size_t d = h->hp.n_audio_state;
    auto half_copy = [&](const float* w, size_t n, const void** out) -> int {
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, n * 2));
      h->derived.push_back(reinterpret_cast<float*>(p));
      h->derived_bytes += n * 2;
      HIP_TRY(convert_f32_to_f16(w, p, (long)n, h->stream));
      *out = p;
      return CRISPY_OK;
    };
    int rc = half_copy(h->conv2_w, d * 3 * d, &h->conv2_wh);
    if (rc == CRISPY_OK) {
      // conv1 kernel [d][3 n_mels] as f16 rows zero-padded to a multiple of 32 columns (3 * 80 = 240 -> 256): the
      // padded operand columns of A read on into the next frames (finite values) and meet zeros here
      const int k1 = 3 * h->hp.n_mels, k1p = (k1 + 31) / 32 * 32;
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, d * k1p * 2));
      h->derived.push_back(reinterpret_cast<float*>(p));
      h->derived_bytes += d * k1p * 2;
      HIP_TRY(hipMemsetAsync(p, 0, d * k1p * 2, h->stream));
      HIP_TRY(convert_rows_f32_to_f16(h->conv1_w, k1, p, k1p, k1, (long)d, h->stream));
      h->conv1_wh = p;
      h->conv1_kp = k1p;
    }
    for (EncLayer& L : h->enc) {
      if (rc == CRISPY_OK) rc = half_copy(L.qkv_w, 3 * d * d, &L.qkv_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.out_w, d * d, &L.out_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.fc1_w, 4 * d * d, &L.fc1_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.fc2_w, 4 * d * d, &L.fc2_wh);
    }
    for (DecLayer& L : h->dec)
    {
      const size_t dtt = h->hp.n_text_state;
      if (rc == CRISPY_OK) rc = half_copy(L.xkv_w, 2 * dtt * dtt, &L.xkv_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.out_w, dtt * dtt, &L.out_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.xout_w, dtt * dtt, &L.xout_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.fc2_w, 4 * dtt * dtt, &L.fc2_wh);
    }
    if (rc != CRISPY_OK) return rc;
    const int dt = h->hp.n_text_state;
    if (dt == 384 || dt == 512 || dt == 768 || dt == 1024 || dt == 1280) {    // the widths the f16 logits kernel is built for
      // token embedding for the logits: f16 (as the model file holds it), packed in MFMA operand order
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, vocab_f16_packed_bytes(h->hp.n_vocab, dt)));
      h->derived.push_back(reinterpret_cast<float*>(p));
      h->derived_bytes += vocab_f16_packed_bytes(h->hp.n_vocab, dt);
      HIP_TRY(pack_vocab_f16(h->tok_emb, p, h->hp.n_vocab, dt, h->stream));
      h->tok_emb_hp = p;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->half_ready = true;
  }
  if (want_ln16 && !h->resident && !h->ln16_ready) {
    // f16 copies of the decoder's q | k | v, cross-q and fc1 weights (un-folded: the LayerNorm runs as a launch of its own)
    const size_t dtt = h->hp.n_text_state;
    auto half_copy = [&](const float* w, size_t n, const void** out) -> int {
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, n * 2));
      h->derived.push_back(reinterpret_cast<float*>(p));
      h->derived_bytes += n * 2;
      HIP_TRY(convert_f32_to_f16(w, p, (long)n, h->stream));
      *out = p;
      return CRISPY_OK;
    };
    int rc = CRISPY_OK;
    for (DecLayer& L : h->dec) {
      if (rc == CRISPY_OK) rc = half_copy(L.qkv_w, 3 * dtt * dtt, &L.qkv_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.xq_w, dtt * dtt, &L.xq_wh);
      if (rc == CRISPY_OK) rc = half_copy(L.fc1_w, 4 * dtt * dtt, &L.fc1_wh);
      if (fused_decode_supported((int)dtt, 1, h->hp.n_audio_ctx)) {      // the fused step kernels' operand order (same bytes once more)
        auto packed = [&](const void* src, size_t n, int kind, const void** out) -> int {
          void* p = nullptr;
          HIP_TRY(hipMalloc(&p, n * 2));
          h->derived.push_back(reinterpret_cast<float*>(p));
          h->derived_bytes += n * 2;
          HIP_TRY(fused_pack_weights(src, p, (int)dtt, kind, h->stream));
          *out = p;
          return CRISPY_OK;
        };
        if (rc == CRISPY_OK) rc = packed(L.qkv_wh, 3 * dtt * dtt, 0, &L.qkv_p);
        if (rc == CRISPY_OK) rc = packed(L.out_wh, dtt * dtt, 1, &L.out_p);
        if (rc == CRISPY_OK) rc = packed(L.fc1_wh, 4 * dtt * dtt, 2, &L.fc1_p);
        if (rc == CRISPY_OK) rc = packed(L.fc2_wh, 4 * dtt * dtt, 3, &L.fc2_p);
      }
    }
    if (rc != CRISPY_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->ln16_ready = true;
  }
  if (h->enc_precision != mode || h->dec_ln16 != want_ln16 || h->dec_attn16 != want_attn16) {   // the captured decode steps bake the kernels in
    h->drop_graphs();
  }
  h->enc_precision = mode;
  h->dec_ln16 = want_ln16;
  h->dec_attn16 = want_attn16;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_set_precision")

int crispy_asr_encode_device(crispy_asr* h, const float* d_mel_t, int batch, float* d_out, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode_device: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_encode_device: model not finalized");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode_device: batch < 0");
  if (batch == 0) return CRISPY_OK;
  if (!d_mel_t || !d_out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode_device: NULL argument");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  int rc = reserve_enc(h, batch);
  if (rc != CRISPY_OK) return rc;
  // A resident quantised model de-quantises every weight into ONE scratch slot in front of its product; decode, cross K|V
  // and the LayerNorm folding fill it on the handle's own stream.  A caller's stream is not ordered against that one, so
  // an encode enqueued here while a decode of the same handle is still in flight would overwrite weights in use
  // (ADVICE r3): order the two explicitly -- this encode starts after everything enqueued on the handle's stream so far,
  // and the handle's stream continues after it.
  struct ScratchOrder {
    crispy_asr* h; hipStream_t s; bool on;
    ~ScratchOrder() {
      if (on && hipEventRecord(h->ev_scratch, s) == hipSuccess) (void)hipStreamWaitEvent(h->stream, h->ev_scratch, 0);
    }
  } scratch_order{h, s, false};
  if (h->resident && s != h->stream) {
    if (!h->ev_scratch) HIP_TRY(hipEventCreateWithFlags(&h->ev_scratch, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->ev_scratch, h->stream));
    HIP_TRY(hipStreamWaitEvent(s, h->ev_scratch, 0));
    scratch_order.on = true;
  }
  const int d = h->hp.n_audio_state, Tn = h->hp.n_audio_ctx, nm = h->hp.n_mels, H = h->hp.n_audio_head;
  const long rows = (long)batch * Tn;
  if (h->enc_precision == 1) {
    // The convolution stem on the f16 matrix cores too (ggml runs a convolution as im2col in f16 x f16 kernel): the
    // frame-major mel is rounded to f16 once, conv1 writes GELU(h1) as f16 (it only feeds conv2), conv2 reads it as
    // a strided view and writes the f32 residual stream.  The f16 buffers alias the hidden-layer / h1 workspace.
    _Float16* mel_h = reinterpret_cast<_Float16*>(h->w_h);                 // [batch][3002][n_mels] (+ padding)
    _Float16* h1_h = reinterpret_cast<_Float16*>(h->w_h1);                 // [batch][3001][d], row 0 of a clip = zeros
    const long mel_n = (long)batch * (MEL_FRAMES + 2) * nm;
    HIP_TRY(convert_f32_to_f16(d_mel_t, mel_h, mel_n, s));
    HIP_TRY(hipMemsetAsync(mel_h + mel_n, 0, 64 * sizeof(_Float16), s));  // what the last row's padded columns read
    HIP_TRY(hipMemset2DAsync(h1_h, (size_t)(MEL_FRAMES + 1) * d * 2, 0, (size_t)d * 2, batch, s));
    {
      HGemmArgs g{};
      g.A = mel_h; g.lda = nm; g.strideA = (long)(MEL_FRAMES + 2) * nm;
      g.W = reinterpret_cast<const _Float16*>(h->conv1_wh); g.ldw = h->conv1_kp;
      g.C = h1_h + d; g.ldc = d; g.strideC = (long)(MEL_FRAMES + 1) * d;
      g.bias = h->conv1_b; g.M = MEL_FRAMES; g.N = d; g.K = h->conv1_kp; g.gelu = 1;
      HIP_TRY(gemm_hh(g, HGEMM_F16, batch, s));
    }
    {
      HGemmArgs g{};
      g.A = h1_h; g.lda = 2L * d; g.strideA = (long)(MEL_FRAMES + 1) * d;
      g.W = reinterpret_cast<const _Float16*>(h->conv2_wh); g.ldw = 3L * d;
      g.C = h->w_x; g.ldc = d; g.strideC = (long)Tn * d;
      g.bias = h->conv2_b; g.M = Tn; g.N = d; g.K = 3 * d;
      g.rowtab = h->enc_pos; g.rowtab_period = Tn;
      HIP_TRY(gemm_hh(g, HGEMM_TAB, batch, s));
    }
  } else {
    // conv1 (k3, p1) + GELU: rows t of the padded frame-major mel are 3*n_mels contiguous floats
    // (the zero row in front of every clip's h1 is rewritten each call: precision mode 1 uses the same workspace as f16)
    HIP_TRY(hipMemset2DAsync(h->w_h1, (size_t)(MEL_FRAMES + 1) * d * sizeof(float), 0, (size_t)d * sizeof(float), batch, s));
    {
      GemmArgs g = gemm(d_mel_t, nm, h->conv1_w, 3L * nm, h->w_h1 + d, d, h->conv1_b, MEL_FRAMES, d, 3 * nm);
      g.strideA = (long)(MEL_FRAMES + 2) * nm;
      g.strideC = (long)(MEL_FRAMES + 1) * d;
      g.gelu = 1;
      HIP_TRY(gemm_f32_nt(g, batch, s));
    }
    // conv2 (k3, s2, p1) + GELU + positional embedding: row t' = frames 2t'-1 .. 2t'+1 of h1 (one zero row in front)
    {
      GemmArgs g = gemm(h->w_h1, 2L * d, h->conv2_w, 3L * d, h->w_x, d, h->conv2_b, Tn, d, 3 * d);
      g.strideA = (long)(MEL_FRAMES + 1) * d;
      g.strideC = (long)Tn * d;
      g.gelu = 1;
      g.rowtab = h->enc_pos;
      g.rowtab_period = Tn;
      HIP_TRY(gemm_f32_nt(g, batch, s));
    }
  }
  if (h->enc_precision == 1) {
    // whisper.cpp's numerics with the bytes halved: every activation that only feeds a matrix product is stored as the
    // f16 value the product would round it to anyway (LayerNorm output, q | k, V^T, attention output, MLP hidden
    // layer); the residual stream stays f32.  The f16 buffers alias the f32 workspace of the default mode.
    void* xn_h = h->w_xn;                                                   // [rows][d] f16
    void* qk_h = h->w_qkv;                                                  // [rows][2 d] f16
    void* vt_h = reinterpret_cast<_Float16*>(h->w_qkv) + rows * 2L * d;     // [batch][d][ENC_TP] f16
    void* att_h = h->w_att;                                                 // [rows][d] f16
    void* hid_h = h->w_h;                                                   // [rows][4 d] f16
    const int swz = h->xcd_swizzle;
    auto hg = [&](const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias, int N, int K) {
      HGemmArgs g{};
      g.A = reinterpret_cast<const _Float16*>(A); g.lda = lda; g.W = reinterpret_cast<const _Float16*>(W); g.ldw = ldw;
      g.C = C; g.ldc = ldc; g.bias = bias; g.M = (int)rows; g.N = N; g.K = K; g.vt_T = Tn; g.xcd_swizzle = swz;
      return g;
    };
    // f16 weights of a product: the resident copy, or -- resident quantised model -- the blocks de-quantised into the
    // scratch slot right here (the previous product has finished with the slot: same stream)
    auto w16 = [&](const void* dense, const QRef& r, const void** out) -> int {
      if (!h->resident) { *out = dense; return CRISPY_OK; }
      return dq(h, r, true, nullptr, s, out);
    };
    for (const EncLayer& L : h->enc) {
      const void* w = nullptr;
      HIP_TRY(layernorm_f16out(h->w_x, L.ln1_w, L.ln1_b, xn_h, rows, d, s));
      if ((rc = w16(L.qkv_wh, L.r_qkv, &w)) != CRISPY_OK) return rc;
      HIP_TRY(gemm_hh(hg(xn_h, d, w, d, qk_h, 2L * d, L.qkv_b, 2 * d, d), HGEMM_F16, 1, s));
      HIP_TRY(gemm_hh(hg(xn_h, d, reinterpret_cast<const _Float16*>(w) + 2L * d * d, d, vt_h, 0, L.qkv_b + 2 * d, d, d),
                      HGEMM_VT, 1, s));
      HIP_TRY(attn_encoder_h(qk_h, vt_h, att_h, batch, Tn, d, H, s, h->dec_attn16 ? 1 : 0));     // mode 2: ggml's rounding points inside the attention
      {
        if ((rc = w16(L.out_wh, L.r_out, &w)) != CRISPY_OK) return rc;
        HGemmArgs g = hg(att_h, d, w, d, h->w_x, d, L.out_b, d, d);
        g.residual = h->w_x; g.ldr = d;
        HIP_TRY(gemm_hh(g, HGEMM_RES, 1, s));
      }
      HIP_TRY(layernorm_f16out(h->w_x, L.ln2_w, L.ln2_b, xn_h, rows, d, s));
      {
        if ((rc = w16(L.fc1_wh, L.r_fc1, &w)) != CRISPY_OK) return rc;
        HGemmArgs g = hg(xn_h, d, w, d, hid_h, 4L * d, L.fc1_b, 4 * d, d);
        g.gelu = 1;
        HIP_TRY(gemm_hh(g, HGEMM_F16, 1, s));
      }
      {
        if ((rc = w16(L.fc2_wh, L.r_fc2, &w)) != CRISPY_OK) return rc;
        HGemmArgs g = hg(hid_h, 4L * d, w, 4L * d, h->w_x, d, L.fc2_b, d, 4 * d);
        g.residual = h->w_x; g.ldr = d;
        HIP_TRY(gemm_hh(g, HGEMM_RES, 1, s));
      }
    }
  } else
  for (const EncLayer& L : h->enc) {
    HIP_TRY(layernorm_f32(h->w_x, L.ln1_w, L.ln1_b, h->w_xn, rows, d, s));
    HIP_TRY(gemm_f32_nt(gemm(h->w_xn, d, L.qkv_w, d, h->w_qkv, 3L * d, L.qkv_b, (int)rows, 3 * d, d), 1, s));
    HIP_TRY(attn_encoder_f32(h->w_qkv, h->w_att, batch, Tn, d, H, s));
    {
      GemmArgs g = gemm(h->w_att, d, L.out_w, d, h->w_x, d, L.out_b, (int)rows, d, d);
      g.residual = h->w_x; g.ldr = d;
      HIP_TRY(gemm_f32_nt(g, 1, s));
    }
    HIP_TRY(layernorm_f32(h->w_x, L.ln2_w, L.ln2_b, h->w_xn, rows, d, s));
    {
      GemmArgs g = gemm(h->w_xn, d, L.fc1_w, d, h->w_h, 4L * d, L.fc1_b, (int)rows, 4 * d, d);
      g.gelu = 1;
      HIP_TRY(gemm_f32_nt(g, 1, s));
    }
    {
      GemmArgs g = gemm(h->w_h, 4L * d, L.fc2_w, 4L * d, h->w_x, d, L.fc2_b, (int)rows, d, 4 * d);
      g.residual = h->w_x; g.ldr = d;
      HIP_TRY(gemm_f32_nt(g, 1, s));
    }
  }
  HIP_TRY(layernorm_f32(h->w_x, h->ln_post_w, h->ln_post_b, d_out, rows, d, s));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_encode_device")

// PCM (host) -> log-mel -> encoder output (host); one call per batch of <= 30 s clips
int crispy_asr_encode(crispy_asr* h, const float* pcm, long pcm_stride, const int* n_samples, int batch, float* out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode: NULL handle");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_encode: model not finalized");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode: batch < 0");
  if (batch == 0) return CRISPY_OK;
  if (!pcm || !n_samples || !out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_encode: NULL argument");
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
  HIP_TRY(hipMemcpyAsync(out, h->w_enc, (size_t)batch * h->hp.n_audio_ctx * h->hp.n_audio_state * sizeof(float),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_encode")

int crispy_asr_synchronize(crispy_asr* h) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_synchronize: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_synchronize")

}  // extern "C"

namespace {

// Row stride of h->d_logits: the vocabulary padded to a multiple of four floats.  n_vocab is odd (51865): with rows V
// apart every clip's row has another 16-byte alignment, the pick kernels split it over their threads differently, and a
// sum over the row (the log-probability of a pick) comes out with other last bits for the same logits -- enough to
// reorder two best-of decoders that sampled the same tokens.
long logits_ld(const crispy_asr* h) { return ((long)h->hp.n_vocab + 3) & ~3L; }

// Decoder workspace for `batch` rows (sequences with a self K|V cache of their own) over `xclips` audio clips (cross K|V;
// 0: one clip per row).  Grows only; growing frees everything and drops the captured steps.
int reserve_dec(crispy_asr* h, int batch, int xclips = 0) {
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
  HIP_TRY(hipMalloc(&h->d_counters, 4 * sizeof(int)));      // position, pick index, ticket of the fused pick, spare
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
  if (fused_decode_supported((int)dt, 1, (int)Tn)) {
    for (int i = 0; i < 3; ++i) {
      HIP_TRY(hipMalloc(&h->d_fx[i], B * dt * sizeof(float)));
      HIP_TRY(hipMalloc(&h->d_fpart[i], (i == 2 ? dt / 32 : dt / 64) * B * dt * sizeof(float)));
    }
  }
  if (gemv_dec_supported((int)dt, 1))
    HIP_TRY(hipMalloc(&h->d_gvpart, (size_t)GEMV_MAX_M * (dt / 64) * XA_PARTS * XA_PART_FLOATS * sizeof(float)));
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
// rows of one group of fallback passes (whole clips x best_of); a grouping choice only -- every row's bits are those of
// its clip decoded alone
constexpr int kLadderRowsMax = 128;

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

// A step of 1 .. GEMV_MAX_M rows of a catalog-width model (768 / 1024 / 1280) in precision mode 1: the projections as
// matrix-vector products with the LayerNorm computed in the consumer (whisper_dec_gemv.hip) -- 8 launches per layer
// instead of 11, spread over N / 8 workgroups instead of N / 32.  Dense f16 copies or resident blocks of ONE ggml type per
// projection; anything else (a mixed file's dense tensors, precision mode 0, more rows, the multi-position prompt) stays on
// the skinny kernels.  CRISPY_ASR_GEMV=0 (developer build) turns it off for the A/B.
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
  const int dt = h->hp.n_text_state, H = h->hp.n_text_head, Tn = h->hp.n_audio_ctx, C = h->hp.n_text_ctx;
  const int* pos_dev = dev_pos ? h->d_counters : nullptr;
  const int clips = batch;
  if (P < 1) P = 1;
  // a generated token (its embedding written by the pick, its position on the device): the fused step kernels
  if (P == 1 && dev_pos && embedded && want_logits && fused_step_ok(h, clips)) return decoder_step_fused(h, clips, s);
  batch = clips * P;                   // rows of this step
  // <= SKINNY_MAX_M clips: the projections run on the skinny kernel (row blocks of 32 clips), which folds the preceding LayerNorm in and writes q and
  // k|v of the self-attention block from one launch (17 launches fewer per step on Whisper-tiny)
  const bool fold = batch <= SKINNY_MAX_M && dt % 128 == 0;
  if (P > 1 && (!fold || dev_pos || embedded))
    return fail(CRISPY_ERR_INVALID_ARG, "decoder_step: a multi-position step needs the folded path and a host position");
  AttnRows self_rows, cross_rows;
  self_rows.group = P; self_rows.key_step = P > 1 ? 1 : 0;
  self_rows.key_off = h->cur_row_off;        // left-padded prompts (decode_ts): every clip's keys start at its own cache row
  const int xg = h->cur_xgroup;              // sequences (rows with a self K|V cache of their own) per audio clip
  const size_t xclips = (size_t)(clips / xg);
  cross_rows.group = P * xg;
  // precision mode 2: q and the normalised probabilities rounded to f16 inside the attentions over the f16 caches
  cross_rows.attn16 = h->dec_attn16 && h->enc_precision == 1 ? 1 : 0;
  // The cross K|V of all layers and clips against the 256 MB Infinity Cache: while it fits, it is what stays cached from
  // step to step (plain loads: 16 tiny clips = 147 MB, 6.8 ms per call against 7.0 non-temporal); beyond that it is a
  // one-pass stream that only evicts the decoder's weights from the L2s, and is requested non-temporally
  // (AttnRows::stream_kv: 64 tiny clips 11.2 -> 10.3 ms, 256 base clips 46.7 -> 44.2 ms).
  // (not in a multi-position prompt step: the P rows of a clip read the same K|V one after the other, and the repeats are
  // served by the Infinity Cache only if the first read allocates there: 2.06 vs 2.18 ms for the prompt of 128 clips)
  static const bool prompt_nt = dev_env("CRISPY_XKV_PROMPT_NT") != nullptr;      // developer A/B (tools/ab_prompt_nt.sh)
  cross_rows.stream_kv = (P == 1 || prompt_nt) && xclips * h->dec.size() * Tn * 2 * dt * (h->enc_precision == 1 ? 2 : 4) > ((size_t)256 << 20) ? 1 : 0;
  if (!embedded) {    // (a fused pick has written the residual stream already)
    if (h->resident)
      HIP_TRY(embed_tokens_q(h->d_tok, h->q_tok_emb->d, h->q_tok_emb->ttype, h->dec_pos, pos, pos_dev, h->d_dx, batch, dt, s, P,
                             h->cur_row_off));
    else
      HIP_TRY(embed_tokens_f32(h->d_tok, h->tok_emb, h->dec_pos, pos, pos_dev, h->d_dx, batch, dt, s, P, h->cur_row_off));
  }
  // resident quantised model: every weight operand is de-quantised into the scratch slot in front of its product --
  // f32 x gamma for the LayerNorm-folded projections (fold_ln's W' = W . diag(gamma), element for element), f16 for the
  // plain ones, plain f32 on the un-folded path of very large batches
  int qrc = CRISPY_OK;
  auto w32 = [&](const float* dense, const QRef& r, const float* gamma) -> const float* {     // (un-folded path only)
    if (!h->resident) return dense;
    const void* o = nullptr;
    const int e = dq(h, r, false, gamma, s, &o);
    if (e != CRISPY_OK) qrc = e;
    return reinterpret_cast<const float*>(o);
  };
  // One projection of the folded path.  Dense model: W = the f32 (gamma-folded) tensor or its f16 copy.  Resident model:
  // the skinny kernel reads the ggml blocks itself and de-quantises in registers (gemm_skinny_q); shapes it has no form
  // for (and dense tensors of a mixed file) go through the scratch slot and the dense kernel.
  auto proj = [&](GemmArgs g, const float* dense32, const void* dense16, const QRef& r, const float* gamma, bool half) -> int {
    g.w_half = half ? 1 : 0;
    if (!h->resident) {
      g.W = half ? reinterpret_cast<const float*>(dense16) : dense32;
      HIP_TRY(gemm_f32_nt(g, 1, s));
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
      HIP_TRY(gemm_skinny_q(g, s));
      return CRISPY_OK;
    }
    const void* o = nullptr;
    const int e = dq(h, r, half, gamma, s, &o);
    if (e != CRISPY_OK) return e;
    g.W = reinterpret_cast<const float*>(o);
    HIP_TRY(gemm_f32_nt(g, 1, s));
    return CRISPY_OK;
  };
  const bool use_gemv = P == 1 && gemv_step_ok(h, batch);
  for (size_t l = 0; l < h->dec.size(); ++l) {
    const DecLayer& L = h->dec[l];
    float* selfkv = h->d_selfkv + l * (size_t)clips * C * 2 * dt;
    const float* xkv = h->d_xkv + l * xclips * Tn * 2 * dt;
    if (use_gemv) {
      _Float16* kvh = reinterpret_cast<_Float16*>(h->d_selfkv) + l * (size_t)clips * C * 2 * dt;
      _Float16* hid = reinterpret_cast<_Float16*>(h->d_dh);                  // GELU'd hidden units as the f16 fc2 multiplies
      auto weights = [&](GemvArgs& a, const void* dense16, const QRef& r) {
        if (!h->resident) { a.w16 = reinterpret_cast<const _Float16*>(dense16); return; }
        for (int i = 0; i < 3; ++i) a.wq[i] = r.t[i < r.n ? i : 0]->d;
        a.wq_type = r.t[0]->ttype;
        a.wq_rows = (int)(r.t[0]->n / (size_t)r.t[0]->cols);
      };
      self_rows.attn16 = h->dec_attn16 ? 1 : 0;
      {
        GemvArgs a{};
        a.x = h->d_dx; a.ldx = dt; a.ln_g = L.ln1_w; a.ln_b = L.ln1_b; weights(a, L.qkv_wh, L.r_qkv); a.bias = L.qkv_b;
        a.out = h->d_dq; a.ldo = dt; a.kv = kvh; a.kv_row_stride = (long)C * 2 * dt; a.pos = pos; a.pos_dev = pos_dev;
        a.M = batch; a.N = 3 * dt; a.K = dt;
        HIP_TRY(gemv_dec(a, GEMV_QKV, s));
      }
      HIP_TRY(attn_decoder_kv16(h->d_dq, dt, kvh, (long)C * 2 * dt, 2L * dt, 64, 0, dt, dev_pos ? 1 : pos + 1, pos_dev, h->d_datt, dt,
                                batch, H, s, h->dec_max_keys, self_rows));
      auto residual_proj = [&](const float* x32, const _Float16* x16, long ldx, const void* dense16, const QRef& r, const float* bias, int K) -> int {
        GemvArgs a{};
        a.x = x32; a.x16 = x16; a.ldx = ldx; weights(a, dense16, r); a.bias = bias;
        a.out = h->d_dx; a.res = h->d_dx; a.ldo = dt; a.M = batch; a.N = dt; a.K = K;
        HIP_TRY(gemv_dec(a, GEMV_RES, s));
        return CRISPY_OK;
      };
      if ((qrc = residual_proj(h->d_datt, nullptr, dt, L.out_wh, L.r_out, L.out_b, dt)) != CRISPY_OK) return qrc;
      if (!cross_rows.attn16 && h->d_gvpart && Tn <= XA_PARTS * 16 * XA_SLOTS * 8) {
        // [LayerNorm -> cross q of a head -> attention over a quarter of the keys] in one launch of heads x XA_PARTS workgroups,
        // the partial soft-maxes merged by the output projection's prologue: two launches where the step had three
        XattnArgs xa{};
        xa.x = h->d_dx; xa.ldx = dt; xa.ln_g = L.lnx_w; xa.ln_b = L.lnx_b; xa.bq = L.xq_b;
        if (L.xq_wh) xa.w16 = reinterpret_cast<const _Float16*>(L.xq_wh);      // (a resident model keeps this one matrix as f16 too: finalize_resident)
        else { xa.wq = L.r_xq.t[0]->d; xa.wq_type = L.r_xq.t[0]->ttype; }
        xa.xkv = reinterpret_cast<const _Float16*>(h->d_xkv_h) + l * xclips * Tn * 2 * dt; xa.clip_stride = (long)Tn * 2 * dt;
        xa.n_keys = Tn; xa.group = xg; xa.part = h->d_gvpart; xa.rows = batch; xa.D = dt;
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
        HIP_TRY(attn_decoder_kv16(h->d_dq, dt, reinterpret_cast<const char*>(h->d_xkv_h) + l * xclips * Tn * 2 * dt * 2,
                                  (long)Tn * 2 * dt, 64, 64L * Tn, 0, (long)Tn * dt, Tn, nullptr, h->d_datt, dt, batch, H, s, 0,
                                  cross_rows));
        if ((qrc = residual_proj(h->d_datt, nullptr, dt, L.xout_wh, L.r_xout, L.xout_b, dt)) != CRISPY_OK) return qrc;
      }
      {
        GemvArgs a{};
        a.x = h->d_dx; a.ldx = dt; a.ln_g = L.ln2_w; a.ln_b = L.ln2_b; weights(a, L.fc1_wh, L.r_fc1); a.bias = L.fc1_b;
        a.out16 = hid; a.ldo = 4L * dt; a.M = batch; a.N = 4 * dt; a.K = dt;
        HIP_TRY(gemv_dec(a, GEMV_GELU16, s));
      }
      if ((qrc = residual_proj(nullptr, hid, 4L * dt, L.fc2_wh, L.r_fc2, L.fc2_b, 4 * dt)) != CRISPY_OK) return qrc;
      continue;
    }
    // causal self-attention against the cache; k | v of this position go straight into the cache row (b, pos)
    float* kv_dst = selfkv + (dev_pos ? 0 : (size_t)pos * 2 * dt);
    // mode 1: the self K|V cache is f16, as whisper.cpp's kv_self is (it aliases the f32 cache: every decode call
    // starts with its own prefill); the projection stores halves, the attention requests all its keys up front
    const bool kv16 = self_kv_half(h, batch);
    _Float16* selfkv_h = reinterpret_cast<_Float16*>(h->d_selfkv) + l * (size_t)clips * C * 2 * dt;
    self_rows.attn16 = kv16 && h->dec_attn16 ? 1 : 0;
    if (fold) {
      // precision mode 2: LayerNorm as a launch of its own, its output rounded to f16 on the way into the f16 matrix cores
      // against f16 weights (ggml's mul_mat arithmetic for these products too); modes 0 / 1: LayerNorm folded in, f32 operands
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
      if ((qrc = proj(g, L.qkv_lw, L.qkv_wh, L.r_qkv, ln16 ? nullptr : L.ln1_w, ln16)) != CRISPY_OK) return qrc;
      if (stage_kv) {
        const size_t esz = kv16 ? 2 : 4;
        void* dst = kv16 ? static_cast<void*>(selfkv_h + (size_t)pos * 2 * dt) : static_cast<void*>(selfkv + (size_t)pos * 2 * dt);
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)C * 2 * dt * esz, h->d_dh, (size_t)P * 2 * dt * esz, (size_t)P * 2 * dt * esz,
                                 (size_t)clips, hipMemcpyDeviceToDevice, s));
      }
    } else {
      HIP_TRY(layernorm_f32(h->d_dx, L.ln1_w, L.ln1_b, h->d_dxn, batch, dt, s));
      const float* qkv_w = w32(L.qkv_w, L.r_qkv, nullptr);
      if (qrc != CRISPY_OK) return qrc;
      HIP_TRY(gemm_f32_nt(gemm(h->d_dxn, dt, qkv_w, dt, h->d_dq, dt, L.qkv_b, batch, dt, dt), 1, s));
      GemmArgs g = gemm(h->d_dxn, dt, qkv_w + (size_t)dt * dt, dt, kv_dst, (long)C * 2 * dt, L.qkv_b + dt, batch, 2 * dt, dt);
      if (dev_pos) { g.c_off_dev = h->d_counters; g.c_off_scale = 2L * dt; }
      HIP_TRY(gemm_f32_nt(g, 1, s));
    }
    if (kv16)
      HIP_TRY(attn_decoder_kv16(h->d_dq, dt, selfkv_h, (long)C * 2 * dt, 2L * dt, 64, 0, dt, dev_pos ? 1 : pos + 1, pos_dev,
                                h->d_datt, dt, batch, H, s, h->dec_max_keys, self_rows));
    else
      HIP_TRY(attn_decoder_f32(h->d_dq, dt, selfkv, (long)C * 2 * dt, 2L * dt, 64, 0, dt, dev_pos ? 1 : pos + 1, pos_dev,
                               h->d_datt, dt, batch, H, s, self_rows));
    // mode 1: the projections that have no LayerNorm in front (attention outputs, the MLP's second GEMM) in ggml's
    // arithmetic -- f16 weights, the f32 activation rounded to f16 on the way into the matrix cores, f32 accumulation
    const bool wh = fold && h->enc_precision == 1 && (L.out_wh || h->resident);
    {
      GemmArgs g = gemm(h->d_datt, dt, L.out_w, dt, h->d_dx, dt, L.out_b, batch, dt, dt);
      g.residual = h->d_dx; g.ldr = dt;
      if ((qrc = proj(g, L.out_w, L.out_wh, L.r_out, nullptr, wh)) != CRISPY_OK) return qrc;
    }
    // cross-attention over the encoder output (K | V precomputed once per clip)
    if (fold) {
      const bool ln16 = h->dec_ln16;
      if (ln16) HIP_TRY(layernorm_f32(h->d_dx, L.lnx_w, L.lnx_b, h->d_dxn, batch, dt, s));
      GemmArgs g = gemm(ln16 ? h->d_dxn : h->d_dx, dt, nullptr, dt, h->d_dq, dt, ln16 ? L.xq_b : nullptr, batch, dt, dt);
      if (!ln16) { g.ln_s = L.xq_ls; g.ln_c = L.xq_lc; }
      if ((qrc = proj(g, L.xq_lw, L.xq_wh, L.r_xq, ln16 ? nullptr : L.lnx_w, ln16)) != CRISPY_OK) return qrc;
    } else {
      HIP_TRY(layernorm_f32(h->d_dx, L.lnx_w, L.lnx_b, h->d_dxn, batch, dt, s));
      const float* xq_w = w32(L.xq_w, L.r_xq, nullptr);
      if (qrc != CRISPY_OK) return qrc;
      HIP_TRY(gemm_f32_nt(gemm(h->d_dxn, dt, xq_w, dt, h->d_dq, dt, L.xq_b, batch, dt, dt), 1, s));
    }
    if (h->enc_precision == 1)
      HIP_TRY(attn_decoder_kv16(h->d_dq, dt, reinterpret_cast<const char*>(h->d_xkv_h) + l * xclips * Tn * 2 * dt * 2,
                                (long)Tn * 2 * dt, 64, 64L * Tn, 0, (long)Tn * dt, Tn, nullptr, h->d_datt, dt, batch, H, s, 0,
                                cross_rows));
    else
      HIP_TRY(attn_decoder_f32(h->d_dq, dt, xkv, (long)Tn * 2 * dt, 64, 64L * Tn, 0, (long)Tn * dt, Tn, nullptr, h->d_datt, dt,
                               batch, H, s, cross_rows));
    {
      GemmArgs g = gemm(h->d_datt, dt, L.xout_w, dt, h->d_dx, dt, L.xout_b, batch, dt, dt);
      g.residual = h->d_dx; g.ldr = dt;
      if ((qrc = proj(g, L.xout_w, L.xout_wh, L.r_xout, nullptr, wh)) != CRISPY_OK) return qrc;
    }
    // MLP
    if (fold) {
      const bool ln16 = h->dec_ln16;
      if (ln16) HIP_TRY(layernorm_f32(h->d_dx, L.ln2_w, L.ln2_b, h->d_dxn, batch, dt, s));
      GemmArgs g = gemm(ln16 ? h->d_dxn : h->d_dx, dt, nullptr, dt, h->d_dh, 4L * dt, ln16 ? L.fc1_b : nullptr, batch, 4 * dt, dt);
      if (!ln16) { g.ln_s = L.fc1_ls; g.ln_c = L.fc1_lc; }
      g.gelu = h->enc_precision == 1 ? 2 : 1;      // mode 1: ggml's GELU (asr_common.h: gelu_ggml)
      if ((qrc = proj(g, L.fc1_lw, L.fc1_wh, L.r_fc1, ln16 ? nullptr : L.ln2_w, ln16)) != CRISPY_OK) return qrc;
    } else {
      HIP_TRY(layernorm_f32(h->d_dx, L.ln2_w, L.ln2_b, h->d_dxn, batch, dt, s));
      GemmArgs g = gemm(h->d_dxn, dt, w32(L.fc1_w, L.r_fc1, nullptr), dt, h->d_dh, 4L * dt, L.fc1_b, batch, 4 * dt, dt);
      if (qrc != CRISPY_OK) return qrc;
      g.gelu = h->enc_precision == 1 ? 2 : 1;
      HIP_TRY(gemm_f32_nt(g, 1, s));
    }
    {
      GemmArgs g = gemm(h->d_dh, 4L * dt, L.fc2_w, 4L * dt, h->d_dx, dt, L.fc2_b, batch, dt, 4 * dt);
      g.residual = h->d_dx; g.ldr = dt;
      if ((qrc = proj(g, L.fc2_w, L.fc2_wh, L.r_fc2, nullptr, wh)) != CRISPY_OK) return qrc;
    }
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
struct Special {
  int sot, lang0, n_lang, n_lang_slots, translate, transcribe, solm, prev, nosp, not_, beg;
  bool multilingual;
};
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
              int* n_out, int xgroup = 1) {
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

// One pass of whisper_full's BEAM_SEARCH strategy over one window per clip [UPSTREAM-RECALL: whisper_full_with_state,
// whisper_sample_token_topk; restated in oracle/whisper_oracle.py: decode_temperature(beam_size=)].  Every clip has n_dec
// decoders (rows [c n_dec, (c + 1) n_dec): beam_size of them at temperature 0, best_of above) over ONE cross K | V.  Per step:
//   * every decoder that is neither completed nor failed DRAWS n_cand ids from its distribution (std::discrete_distribution
//     over the probabilities the rules leave at this temperature, n_cand variates from the decoder's own generator -- the
//     device pick kernel in its candidate form) -> candidates (decoder, sequence + id, sum of ALL log-probabilities);
//   * the clip's candidates are sorted by that sum (descending; ties: decoder index) and dealt to the live decoders in
//     order, skipping candidates whose token sequence equals the one just dealt (not at the first step); a decoder takes
//     the candidate's sequence, window state and -- on the device -- the self K | V rows of the decoder it came from;
//   * completion / failure bookkeeping as in the sampling pass; the next decoder step feeds every live row its last id.
// The host decides between steps (one round trip per token: this is the strategy's structure, not a captured loop).
// rng[r]: the generator of row r's decoder; it advances by n_cand variates per step the decoder is live.
// Outputs as decode_ts: the sequence every decoder ENDS with.
int decode_beam(crispy_asr* h, const float* d_enc, int n_clips, int n_dec, int n_cand, const std::vector<std::vector<int>>& clip_prompts,
                int rules, const int* seek, const int* seek_end, int max_new, const unsigned char* mask, const unsigned char* mask_first,
                float temperature, const std::vector<std::mt19937*>& rng, int* tokens_out, int* tids_out, float* plog_out,
                float* nosp_out, int* n_out) {
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int rows = n_clips * n_dec;
  if (n_clips < 1 || n_dec < 1 || n_dec > TS_MAX_CAND || n_cand < 1 || n_cand > TS_MAX_CAND || (int)clip_prompts.size() != n_clips ||
      (int)rng.size() != rows)
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
  HIP_TRY(hipMemcpyAsync(h->d_temperature, &t_eff, sizeof(float), hipMemcpyHostToDevice, s));
  // the bytes of a row's cache the decoders of a clip can differ in: the generated positions
  const size_t esz = self_kv_half(h, rows) ? 2 : 4;
  const size_t row_bytes = (size_t)C * 2 * dt * esz, pos_bytes = (size_t)2 * dt * esz;
  const size_t need = (size_t)L * rows * (size_t)max_new * pos_bytes;
  if (need > h->beam_kv_bytes) {
    if (h->d_beam_kv) (void)hipFree(h->d_beam_kv);
    h->d_beam_kv = nullptr; h->beam_kv_bytes = 0;
    HIP_TRY(hipMalloc(&h->d_beam_kv, need));
    h->beam_kv_bytes = need;
  }
  struct Seq {
    std::vector<int> toks, tids;
    std::vector<float> plog;
    double sum_all = 0.0;
    bool has_ts = false, failed = false, completed = false;
    int seek_delta = 3000, result_len = 0;
    TsState st;
  };
  std::vector<Seq> seq((size_t)rows);
  for (int r = 0; r < rows; ++r) {
    const int c = r / n_dec;
    seq[r].st = TsState{-1, -1, 0, -1, 0, seek ? seek[c] : 0, seek_end ? seek_end[c] : (1 << 30), 0};
  }
  TsPickArgs pa = ts_args(h, rules, mask, mask_first);
  pa.u_all = h->d_u_all;
  pa.n_cand = n_cand;
  pa.cand_tok = h->d_tokens_all; pa.cand_plog = h->d_plog_all; pa.cand_tid = h->d_tids_all;
  const int delta_min = TS_DELTA_MIN;
  std::vector<double> u((size_t)rows * n_cand);
  std::vector<TsState> st((size_t)rows);
  std::vector<int> c_tok((size_t)rows * n_cand), c_tid((size_t)rows * n_cand), parent((size_t)rows), feed((size_t)rows);
  std::vector<float> c_plog((size_t)rows * n_cand);
  struct Cand { int j, k; double sum; };
  for (int i = 0; i < max_new; ++i) {
    for (int r = 0; r < rows; ++r) {
      const bool live = !(seq[r].completed || seq[r].failed);
      for (int k = 0; k < n_cand; ++k) u[(size_t)r * n_cand + k] = live ? canonical(*rng[r]) : 0.5;
      st[r] = seq[r].st;
      st[r].done = live ? 0 : 1;
    }
    HIP_TRY(hipMemcpyAsync(h->d_u_all, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_ts_state, st.data(), st.size() * sizeof(TsState), hipMemcpyHostToDevice, s));
    HIP_TRY(ts_pick(pa, rows, s));
    HIP_TRY(hipMemcpyAsync(c_tok.data(), h->d_tokens_all, c_tok.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(c_tid.data(), h->d_tids_all, c_tid.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(c_plog.data(), h->d_plog_all, c_plog.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    bool moved = false, any_live = false;
    for (int r = 0; r < rows; ++r) parent[r] = r;
    for (int c = 0; c < n_clips; ++c) {
      const int r0 = c * n_dec;
      std::vector<Cand> cands;
      for (int j = 0; j < n_dec; ++j) {
        const Seq& q = seq[r0 + j];
        if (q.completed || q.failed) continue;
        for (int k = 0; k < n_cand; ++k) {
          const size_t x = (size_t)(r0 + j) * n_cand + k;
          if (c_tok[x] < 0 || c_tok[x] >= h->hp.n_vocab) return fail(CRISPY_ERR_HIP, "beam decode: candidate id %d", c_tok[x]);
          cands.push_back(Cand{j, k, q.sum_all + (double)c_plog[x]});
        }
      }
      if (cands.empty()) continue;
      std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) {
        if (a.sum != b.sum) return a.sum > b.sum;
        return a.j < b.j;
      });
      auto tok_of = [&](const Cand& x) { return c_tok[(size_t)(r0 + x.j) * n_cand + x.k]; };
      auto same = [&](const Cand& a, const Cand& b) {      // whisper_sequence_tokens_equal of the two candidates' sequences
        return tok_of(a) == tok_of(b) && (a.j == b.j || seq[r0 + a.j].toks == seq[r0 + b.j].toks);
      };
      std::vector<Seq> next(seq.begin() + r0, seq.begin() + r0 + n_dec);
      size_t cur_c = 0;
      for (int j = 0; j < n_dec; ++j) {
        if (seq[r0 + j].completed || seq[r0 + j].failed) continue;
        if (cur_c >= cands.size()) cur_c = 0;
        const Cand cur = cands[cur_c++];
        while (cands.size() > cur_c && i > 0 && same(cands[cur_c], cur)) ++cur_c;
        const size_t x = (size_t)(r0 + cur.j) * n_cand + cur.k;
        Seq q = seq[r0 + cur.j];
        q.toks.push_back(c_tok[x]); q.tids.push_back(c_tid[x]); q.plog.push_back(c_plog[x]);
        q.sum_all = cur.sum;
        // the rules' view of the sequence (the pick kernel's ts_commit)
        q.st.prev = q.st.last; q.st.last = c_tok[x]; q.st.n += 1;
        if (rules == TS_RULES_OPENAI ? c_tok[x] >= sp.beg : c_tok[x] > sp.beg) q.st.last_ts = c_tok[x];
        next[j] = std::move(q);
        parent[r0 + j] = r0 + cur.j;
        moved = moved || cur.j != j;
      }
      std::move(next.begin(), next.end(), seq.begin() + r0);
      // completion / failure of every live decoder on its new last token
      for (int j = 0; j < n_dec; ++j) {
        Seq& d = seq[r0 + j];
        if (d.completed || d.failed) continue;
        const int t = d.toks.back();
        const int sk = d.st.seek, se = d.st.seek_end;
        if (t > sp.beg) {
          const int sd = 2 * (t - sp.beg);
          if (d.has_ts && d.seek_delta > sd && d.result_len < i) { d.failed = true; continue; }      // "do not allow to go back in time"
          d.seek_delta = sd; d.result_len = i + 1; d.has_ts = true;
        }
        if (t == h->eot || (d.has_ts && sk + d.seek_delta + delta_min >= se)) {
          if (d.result_len == 0) {
            if (sk + d.seek_delta + delta_min >= se) d.result_len = i + 1;
            else { d.failed = true; continue; }
          }
          d.completed = true;
          continue;
        }
        if (i == max_new - 1 && (d.result_len == 0 || d.seek_delta < 1500)) { d.failed = true; continue; }
        any_live = true;
      }
    }
    if (!any_live || i == max_new - 1) break;
    if (moved && i > 0) {
      HIP_TRY(hipMemcpyAsync(h->d_beam_parent, parent.data(), sizeof(int) * rows, hipMemcpyHostToDevice, s));
      HIP_TRY(beam_kv_reorder(h->d_selfkv, h->d_beam_kv, h->d_beam_parent, L, rows, (long)row_bytes, (long)((size_t)pos * pos_bytes),
                              (long)((size_t)i * pos_bytes), s));
    }
    for (int r = 0; r < rows; ++r) feed[r] = (seq[r].completed || seq[r].failed || seq[r].toks.empty()) ? h->eot : seq[r].toks.back();
    HIP_TRY(hipMemcpyAsync(h->d_tok, feed.data(), sizeof(int) * rows, hipMemcpyHostToDevice, s));
    rc = decoder_step(h, rows, pos + i, false, true, s);
    if (rc != CRISPY_OK) return rc;
  }
  std::vector<float> nosp(rows);
  HIP_TRY(hipMemcpyAsync(nosp.data(), h->d_nosp, nosp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int r = 0; r < rows; ++r) {
    const Seq& q = seq[r];
    const int n = std::min<int>((int)q.toks.size(), max_new);
    for (int i = 0; i < max_new; ++i) {
      tokens_out[(size_t)r * max_new + i] = i < n ? q.toks[i] : h->eot;
      if (tids_out) tids_out[(size_t)r * max_new + i] = i < n ? q.tids[i] : sp.beg;
      if (plog_out) plog_out[(size_t)r * max_new + i] = i < n ? q.plog[i] : 0.f;
    }
    if (nosp_out) nosp_out[r] = nosp[r];
    if (n_out) n_out[r] = n;
  }
  return CRISPY_OK;
}

// pick a token from the current logits (step-aware suppression), record it, run the next step on it,
// advance the device counters: the body of one generated token
int generation_body(crispy_asr* h, int batch, hipStream_t s) {
  const StepFuse f = step_fuse(h);         // pick + embedding of the pick + counters in one launch
  HIP_TRY(argmax_f32(h->d_logits, h->d_suppress, h->d_suppress_first, h->d_counters + 1, h->hp.n_vocab, logits_ld(h), h->d_tok,
                     h->d_tokens_all, h->d_best, batch, s, h->eot, h->d_finished, h->d_done_count, &f));
  return decoder_step(h, batch, 0, true, true, s, true);
}

}  // namespace

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

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// whisper.cpp GGML model file [UPSTREAM-RECALL]:
//   u32 magic 0x67676d6c | 11 x i32 hparams (.., n_mels, ftype) | i32 n_mel, i32 n_fft, f32 filters
//   | i32 n_tokens, then (u32 len, bytes) per token | tensors until EOF:
//   i32 n_dims, i32 name_len, i32 ttype, i32 ne[n_dims] (innermost first), name, data.
// f32 / f16 tensors are taken as is; q4_0, q4_1, q5_0, q5_1, q8_0 blocks (the catalog's medium-q4_1 and
// large-v3-q5_0 files, managers/model.rs:99,137) are de-quantised to f32 at load time.
// ---------------------------------------------------------------------------------------------
namespace {

struct FileReader {
  FILE* f = nullptr;
  ~FileReader() { if (f) fclose(f); }
  bool read(void* dst, size_t n) { return fread(dst, 1, n, f) == n; }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      exp = 127 - 15 + 1;
      while (!(man & 0x400)) { man <<= 1; --exp; }
      bits = sign | (exp << 23) | ((man & 0x3ff) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  float out;
  std::memcpy(&out, &bits, 4);
  return out;
}


// ggml block-quantised rows -> f32 [UPSTREAM-RECALL ggml-quants]: blocks of 32 weights along the innermost
// dimension; d (and m) are f16; low nibbles are elements 0..15 of the block, high nibbles 16..31; q5 adds a
// fifth bit per element from the 32-bit mask qh.
struct QuantInfo { int block_bytes; };
bool quant_info(int ttype, QuantInfo* qi) {
  switch (ttype) {
    case 2: qi->block_bytes = 2 + 16; return true;           // q4_0
    case 3: qi->block_bytes = 2 + 2 + 16; return true;       // q4_1
    case 6: qi->block_bytes = 2 + 4 + 16; return true;       // q5_0
    case 7: qi->block_bytes = 2 + 2 + 4 + 16; return true;   // q5_1
    case 8: qi->block_bytes = 2 + 32; return true;           // q8_0
    default: return false;
  }
}
void dequant_block(int ttype, const uint8_t* b, float* y) {
  auto h = [&](const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return half_to_float(v); };
  if (ttype == 2) {
    const float d = h(b); const uint8_t* qs = b + 2;
    for (int j = 0; j < 16; ++j) { y[j] = ((qs[j] & 0x0F) - 8) * d; y[j + 16] = ((qs[j] >> 4) - 8) * d; }
  } else if (ttype == 3) {
    const float d = h(b), m = h(b + 2); const uint8_t* qs = b + 4;
    for (int j = 0; j < 16; ++j) { y[j] = (qs[j] & 0x0F) * d + m; y[j + 16] = (qs[j] >> 4) * d + m; }
  } else if (ttype == 6) {
    const float d = h(b); uint32_t qh; std::memcpy(&qh, b + 2, 4); const uint8_t* qs = b + 6;
    for (int j = 0; j < 16; ++j) {
      const int x0 = (qs[j] & 0x0F) | (((qh >> j) & 1) << 4);
      const int x1 = (qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      y[j] = (x0 - 16) * d; y[j + 16] = (x1 - 16) * d;
    }
  } else if (ttype == 7) {
    const float d = h(b), m = h(b + 2); uint32_t qh; std::memcpy(&qh, b + 4, 4); const uint8_t* qs = b + 8;
    for (int j = 0; j < 16; ++j) {
      const int x0 = (qs[j] & 0x0F) | (((qh >> j) & 1) << 4);
      const int x1 = (qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      y[j] = x0 * d + m; y[j + 16] = x1 * d + m;
    }
  } else {  // q8_0
    const float d = h(b); const int8_t* qs = reinterpret_cast<const int8_t*>(b + 2);
    for (int j = 0; j < 32; ++j) y[j] = qs[j] * d;
  }
}

}  // namespace

namespace {

struct crispy_asr_result_impl {
  crispy_asr_result pub;
  std::string text;
  std::vector<int> tokens;
  int language_token = 0;
  std::vector<std::string> seg_text;
  std::vector<float> seg_t0, seg_t1;
  std::vector<crispy_asr_segment> segs;
  std::vector<crispy_asr_window> wins;
};

// whisper.cpp's always-suppressed specials (whisper_process_logits [UPSTREAM-RECALL]): sot, nosp, translate,
// transcribe, prev, solm, every language token; suppress_blank adds " " and EOT at the first position.
int build_ts_masks(crispy_asr* h) {
  const Special sp = special_tokens(h);
  const int V = h->hp.n_vocab;
  std::vector<unsigned char> m(V, 0);
  for (int t : {sp.sot, sp.nosp, sp.translate, sp.transcribe, sp.prev, sp.solm})
    if (t >= 0 && t < V) m[t] = 1;
  for (int t = sp.lang0; t < sp.lang0 + sp.n_lang_slots && t < V; ++t) m[t] = 1;
  std::vector<unsigned char> f = m;
  int blank = 220;                                  // " " in both GPT-2 vocabularies
  for (size_t t = 0; t < h->vocab.size(); ++t)
    if (h->vocab[t] == " ") { blank = (int)t; break; }
  if (blank < V) f[blank] = 1;
  if (h->eot < V) f[h->eot] = 1;
  if (!h->d_ts_mask) HIP_TRY(hipMalloc(&h->d_ts_mask, V));
  if (!h->d_ts_mask_first) HIP_TRY(hipMalloc(&h->d_ts_mask_first, V));
  HIP_TRY(hipMemcpy(h->d_ts_mask, m.data(), V, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ts_mask_first, f.data(), V, hipMemcpyHostToDevice));
  return CRISPY_OK;
}

// whisper_full_params.suppress_nst [UPSTREAM-RECALL: whisper.cpp `non_speech_tokens` + whisper_process_logits]: every
// string of the list, as it stands and with a leading space, that the vocabulary holds as ONE token; then " -" and " '"
// ("allow hyphens and single quotes between words, but not at the beginning of a word").  Oracle: whisper_oracle.py
// non_speech_token_ids.
std::vector<int> non_speech_token_ids(const std::vector<std::string>& vocab) {
  static const char* const kList[] = {
      "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
      "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f",          // the four CJK corner brackets
      "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}",
      "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa",      // two / three eighth notes
      "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};
  std::map<std::string, int> id;
  for (size_t t = 0; t < vocab.size(); ++t) id.emplace(vocab[t], (int)t);      // first id of a string, as token_to_id would hold one
  std::vector<int> out;
  auto add = [&](const std::string& s) { auto it = id.find(s); if (it != id.end()) out.push_back(it->second); };
  for (const char* t : kList) { add(t); add(std::string(" ") + t); }
  add(" -");
  add(" '");
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
  return out;
}

int build_nst_masks(crispy_asr* h) {
  if (h->d_ts_mask_nst) return CRISPY_OK;
  if (!h->d_ts_mask) { const int rc = build_ts_masks(h); if (rc != CRISPY_OK) return rc; }
  const int V = h->hp.n_vocab;
  std::vector<unsigned char> m(V), f(V);
  HIP_TRY(hipMemcpy(m.data(), h->d_ts_mask, V, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(f.data(), h->d_ts_mask_first, V, hipMemcpyDeviceToHost));
  for (int t : non_speech_token_ids(h->vocab))
    if (t < V) { m[t] = 1; f[t] = 1; }
  HIP_TRY(hipMalloc(&h->d_ts_mask_nst, V));
  HIP_TRY(hipMalloc(&h->d_ts_mask_first_nst, V));
  HIP_TRY(hipMemcpy(h->d_ts_mask_nst, m.data(), V, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ts_mask_first_nst, f.data(), V, hipMemcpyHostToDevice));
  return CRISPY_OK;
}

// One decoder of one pass over a window: the picks the device made, and whisper_full's bookkeeping replayed over them
// (oracle/whisper_oracle.py: decode_temperature) [UPSTREAM-RECALL: whisper_full_with_state, "update the decoder state"].
struct DecoderPass {
  const int* toks = nullptr;
  const int* tids = nullptr;
  const float* plog = nullptr;
  int n = 0;                         // picks made (the device stops a row at EOT or at a timestamp delta_min from the end)
  bool has_ts = false, failed = false, completed = false, scored = false;
  int seek_delta = 3000, result_len = 0;
  double sum_logprobs = 0, avg_logprobs = -INFINITY, score = -INFINITY, entropy = 0;
};

void replay_decoder(DecoderPass& d, int n_max, int beg, int eot, int seek, int seek_end, int delta_min) {
  for (int i = 0; i < d.n; ++i) {
    const int t = d.toks[i];
    if (t > beg) {
      const int sd = 2 * (t - beg);
      if (d.has_ts && d.seek_delta > sd && d.result_len < i) { d.failed = true; return; }   // "do not allow to go back in time"
      d.seek_delta = sd; d.result_len = i + 1; d.has_ts = true;
    }
    if (t == eot || (d.has_ts && seek + d.seek_delta + delta_min >= seek_end)) {
      if (d.result_len == 0) {
        if (seek + d.seek_delta + delta_min >= seek_end) d.result_len = i + 1;
        else { d.failed = true; return; }                   // end of text before any timestamp: nothing to keep
      }
      d.completed = true;
      return;
    }
    if (i == n_max - 1 && (d.result_len == 0 || d.seek_delta < 1500)) { d.failed = true; return; }   // repetition loop
  }
}

// whisper_sequence_score over the kept tokens: sum / mean log-probability, the ranking score (length_penalty -1: the
// mean), entropy of the token histogram of the last 32
void score_decoder(DecoderPass& d) {
  if (d.result_len == 0) return;
  double sum = 0;
  for (int i = 0; i < d.result_len; ++i) sum += d.plog[i];
  d.sum_logprobs = sum;
  d.avg_logprobs = sum / d.result_len;
  d.score = sum / d.result_len;
  std::map<int, int> cnt;
  int c = 0;
  for (int i = std::max(0, d.result_len - 32); i < d.result_len; ++i) { cnt[d.toks[i]]++; ++c; }
  double ent = 0;
  for (const auto& kv : cnt) {
    const double p = kv.second / (double)c;
    ent -= p * std::log(p);
  }
  d.entropy = ent;
  d.scored = true;
}

// segments of one window as whisper_full builds them (oracle: window_segments); times in seconds
void window_segments(const crispy_asr* h, const int* toks, const int* tids, int n, int beg, int seek, int seek_delta,
                     crispy_asr_result_impl* r) {
  if (n <= 0) return;
  auto piece = [&](int t) -> std::string { return t < (int)h->vocab.size() ? h->vocab[t] : std::string(); };
  int t0 = seek + 2 * (tids[0] - beg);
  std::string text;
  for (int i = 0; i < n; ++i) {
    if (toks[i] < h->eot) text += piece(toks[i]);
    if (toks[i] > beg) {
      const int t1 = seek + 2 * (tids[i] - beg);
      if (!text.empty()) { r->seg_t0.push_back(t0 / 100.f); r->seg_t1.push_back(t1 / 100.f); r->seg_text.push_back(text); }
      text.clear();
      while (i < n && toks[i] > beg) ++i;
      --i;
      t0 = t1;
    }
  }
  if (!text.empty()) { r->seg_t0.push_back(t0 / 100.f); r->seg_t1.push_back((seek + seek_delta) / 100.f); r->seg_text.push_back(text); }
}

void publish(crispy_asr_result_impl* r) {
  r->segs.resize(r->seg_text.size());
  for (size_t i = 0; i < r->segs.size(); ++i) r->segs[i] = crispy_asr_segment{r->seg_t0[i], r->seg_t1[i], r->seg_text[i].c_str()};
  r->pub.text = r->text.c_str();
  r->pub.tokens = r->tokens.data();
  r->pub.n_tokens = (int)r->tokens.size();
  r->pub.language_token = r->language_token;
  r->pub.n_segments = (int)r->segs.size();
  r->pub.segments = r->segs.empty() ? nullptr : r->segs.data();
  r->pub.n_windows = (int)r->wins.size();
  r->pub.windows = r->wins.empty() ? nullptr : r->wins.data();
}

}  // namespace

extern "C" {

namespace {
int load_impl(const char* model_path, int device, bool resident, crispy_asr** out);
}
int crispy_asr_load(const char* model_path, int device, crispy_asr** out) try {
  return load_impl(model_path, device, false, out);
} CRISPY_CATCH_RET("crispy_asr_load")

int crispy_asr_load_resident(const char* model_path, int device, crispy_asr** out) try {
  return load_impl(model_path, device, true, out);
} CRISPY_CATCH_RET("crispy_asr_load_resident")

int crispy_asr_memory_info(const crispy_asr* h, size_t* weight_bytes, size_t* quantised_bytes, size_t* scratch_bytes) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_memory_info: NULL handle");
  size_t dense = h->derived_bytes, q = 0;
  for (const auto& kv : h->tensors)
    if (kv.second.d) dense += kv.second.n * sizeof(float);
  for (const auto& kv : h->qtensors)
    if (kv.second.owned) q += kv.second.nbytes;
  if (weight_bytes) *weight_bytes = dense + q;
  if (quantised_bytes) *quantised_bytes = q;
  if (scratch_bytes) *scratch_bytes = h->q_scratch_bytes;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_memory_info")

namespace {
// the matrices finalize_resident consumes as ggml blocks (QRef): attention and MLP weights, the token embedding.  Any
// other 2-D tensor a file may hold quantised (whisper.cpp's own tool leaves them alone, the format does not forbid it:
// positional embeddings, the [d, 1] convolution biases) is read through T() as dense f32 and is inflated at load.
bool resident_block_name(const std::string& name) {
  if (name == "decoder.token_embedding.weight") return true;
  static const char* const tails[] = {".attn.query.weight", ".attn.key.weight", ".attn.value.weight", ".attn.out.weight",
                                      ".cross_attn.query.weight", ".cross_attn.key.weight", ".cross_attn.value.weight",
                                      ".cross_attn.out.weight", ".mlp.0.weight", ".mlp.2.weight"};
  for (const char* t : tails) {
    const size_t n = std::strlen(t);
    if (name.size() >= n && name.compare(name.size() - n, n, t) == 0) return true;
  }
  return false;
}

int load_impl(const char* model_path, int device, bool resident, crispy_asr** out) {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_load: out is NULL");
  *out = nullptr;
  if (!model_path) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_load: NULL path");
  FileReader r;
  r.f = fopen(model_path, "rb");
  if (!r.f) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: cannot open '%s'", model_path);
  uint32_t magic = 0;
  int32_t hpv[11];
  if (!r.read(&magic, 4) || magic != 0x67676d6c)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: '%s' is not a ggml whisper model (bad magic)", model_path);
  if (!r.read(hpv, sizeof(hpv))) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated header");
  crispy_asr_hparams hp;
  hp.n_vocab = hpv[0]; hp.n_audio_ctx = hpv[1]; hp.n_audio_state = hpv[2]; hp.n_audio_head = hpv[3];
  hp.n_audio_layer = hpv[4]; hp.n_text_ctx = hpv[5]; hp.n_text_state = hpv[6]; hp.n_text_head = hpv[7];
  hp.n_text_layer = hpv[8]; hp.n_mels = hpv[9];
  int32_t fm = 0, ff = 0;
  if (!r.read(&fm, 4) || !r.read(&ff, 4) || fm != hp.n_mels || ff != MEL_BINS)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: mel filter block is %d x %d, expected %d x %d", fm, ff, hp.n_mels,
                MEL_BINS);
  std::vector<float> filters((size_t)fm * ff);
  if (!r.read(filters.data(), filters.size() * 4)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated filters");
  if (hp.n_vocab <= 0 || hp.n_vocab > 65536)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: n_vocab %d is not a whisper vocabulary size", hp.n_vocab);
  int32_t n_tok = 0;
  if (!r.read(&n_tok, 4) || n_tok < 0 || n_tok > hp.n_vocab + 1024)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: bad vocabulary size %d", n_tok);
  std::vector<std::string> vocab(n_tok);
  for (int i = 0; i < n_tok; ++i) {
    uint32_t len = 0;
    if (!r.read(&len, 4) || len > 4096) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: bad token %d", i);
    vocab[i].resize(len);
    if (len && !r.read(&vocab[i][0], len)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated vocabulary");
  }
  crispy_asr* h = nullptr;
  int rc = crispy_asr_create(&hp, filters.data(), device, &h);
  if (rc != CRISPY_OK) return rc;
  h->vocab = std::move(vocab);
  // (h->resident is decided after the tensor loop: only a file that HAS quantised matrices takes the resident path)
  auto bail = [&](int code) {
    const std::string keep = last_error_cstr();
    crispy_asr_free(h);
    return fail(code, "%s", keep.c_str());
  };
  std::vector<float> buf;
  std::vector<uint16_t> hbuf;
  std::vector<uint8_t> qbuf;
  // a tensor is read only if the model needs it and the file's shape has exactly the element count the
  // hyper-parameters imply: the buffers below are sized from this table, never from numbers a corrupt file supplies
  const std::map<std::string, size_t> expect = expected_tensors(hp);
  for (;;) {
    int32_t n_dims = 0, name_len = 0, ttype = 0;
    if (!r.read(&n_dims, 4)) break;  // clean EOF
    if (!r.read(&name_len, 4) || !r.read(&ttype, 4) || n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 256) {
      fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor header");
      return bail(CRISPY_ERR_BAD_MODEL);
    }
    int32_t ne[4] = {1, 1, 1, 1};
    unsigned long long n64 = 1;          // <= (2^31)^4 would overflow: checked against 2^40 after every factor
    for (int i = 0; i < n_dims; ++i) {
      if (!r.read(&ne[i], 4) || ne[i] <= 0) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor shape"); return bail(CRISPY_ERR_BAD_MODEL); }
      n64 *= (unsigned long long)ne[i];
      if (n64 > (1ull << 40)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor shape (element count overflows)"); return bail(CRISPY_ERR_BAD_MODEL); }
    }
    const size_t n = (size_t)n64;
    std::string name(name_len, '\0');
    if (!r.read(&name[0], name_len)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated tensor name"); return bail(CRISPY_ERR_BAD_MODEL); }
    {
      const auto it = expect.find(name);
      if (it == expect.end()) {
        fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: unknown tensor '%s' in model file", name.c_str());
        return bail(CRISPY_ERR_BAD_MODEL);
      }
      if (it->second != n) {
        fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: tensor '%s' has %zu elements, the hyper-parameters imply %zu",
             name.c_str(), n, it->second);
        return bail(CRISPY_ERR_BAD_MODEL);
      }
    }
    buf.resize(n);
    if (ttype == 0) {
      if (!r.read(buf.data(), n * 4)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
    } else if (ttype == 1) {
      hbuf.resize(n);
      if (!r.read(hbuf.data(), n * 2)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
      for (size_t i = 0; i < n; ++i) buf[i] = half_to_float(hbuf[i]);
    } else {
      QuantInfo qi;
      if (!quant_info(ttype, &qi) || ne[0] % 32 != 0) {
        fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_load: tensor '%s' has ggml type %d (supported: f32 0, f16 1, q4_0 2, "
             "q4_1 3, q5_0 6, q5_1 7, q8_0 8; rows must be multiples of 32)", name.c_str(), ttype);
        return bail(CRISPY_ERR_UNSUPPORTED);
      }
      const size_t n_blocks = n / 32;
      qbuf.resize(n_blocks * qi.block_bytes);
      if (!r.read(qbuf.data(), qbuf.size())) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
      if (resident && n_dims == 2 && resident_block_name(name)) {
        // the blocks stay as they are (managers/model.rs:99,137: the catalog's q4_1 / q5_0 files): no f32 tensor is made
        QTensor q;
        q.ttype = ttype; q.n = n; q.cols = ne[0]; q.nbytes = qbuf.size();
        if (hipSetDevice(device) != hipSuccess || hipMalloc(&q.d, q.nbytes + 16) != hipSuccess ||      // (+16: the in-register block fetch reads whole dwords)
            hipMemcpy(q.d, qbuf.data(), q.nbytes, hipMemcpyHostToDevice) != hipSuccess) {
          if (q.d) (void)hipFree(q.d);
          fail(CRISPY_ERR_OOM, "crispy_asr_load_resident: no device memory for '%s' (%zu bytes)", name.c_str(), q.nbytes);
          return bail(CRISPY_ERR_OOM);
        }
        h->qtensors[name] = q;
        h->tensors[name].set = true;
        continue;
      }
      for (size_t bi = 0; bi < n_blocks; ++bi) dequant_block(ttype, qbuf.data() + bi * qi.block_bytes, buf.data() + bi * 32);
    }
    rc = crispy_asr_set_tensor(h, name.c_str(), buf.data(), n);
    if (rc != CRISPY_OK) return bail(rc);
  }
  // A file without a single quantised matrix (f32 / f16: the catalog's ggml-small.bin and large-v3-turbo,
  // managers/model.rs:80,118) loads exactly as crispy_asr_load does: dense tensors, the ordinary finalize, the f16 copies
  // of precision mode 1 -- not the resident path, where every matrix would be copied into the scratch slot in front of
  // every product.
  h->resident = resident && !h->qtensors.empty();
  rc = crispy_asr_finalize(h);
  if (rc != CRISPY_OK) return bail(rc);
  if (resident && !h->resident) {              // what crispy_asr_load_resident promises: whisper.cpp's arithmetic
    rc = crispy_asr_set_precision(h, 1);
    if (rc != CRISPY_OK) return bail(rc);
  }
  *out = h;
  return CRISPY_OK;
}
}  // namespace

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

int crispy_asr_transcribe(crispy_asr* h, const float* pcm16k, size_t n, const crispy_asr_opts* opts,
                          crispy_asr_result** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe: out is NULL");
  *out = nullptr;
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe: NULL handle");
  if (n > 0 && !pcm16k) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe: NULL audio");
  return crispy_asr_transcribe_batch(h, &pcm16k, &n, 1, opts, out);
} CRISPY_CATCH_RET("crispy_asr_transcribe")

// engine.transcribe for a batch of chunks at once; results[i] is library-owned (crispy_asr_free_result each).
//   no_timestamps = 1: prompt [sot, lang, task, <|notimestamps|>], one window, plain greedy arg-max.
//   no_timestamps = 0 (whisper.cpp's default, what TranscribeOptions::default() runs): whisper_full's seek loop
//     [UPSTREAM-RECALL] -- windows of 30 s starting at `seek`, greedy picks under the timestamp rules, the window
//     advances to the last closed timestamp pair, segments are cut at timestamp tokens; per window the no-speech rule and
//     the temperature ladder (best_of sampling decoders above temperature 0) decide what is kept.  Not reproduced: beam
//     search at temperature 0.
int crispy_asr_transcribe_batch(crispy_asr* h, const float* const* pcm, const size_t* n, int batch,
                                const crispy_asr_opts* opts, crispy_asr_result** results) try {
  return transcribe_batch_impl(h, pcm, n, batch, opts, results, nullptr);
} CRISPY_CATCH_RET("crispy_asr_transcribe_batch")

}  // extern "C"

namespace {

// cancel (nullable): polled at the top of every round of the seek loop -- a set flag ends the call with
// CRISPY_ERR_CANCELLED and no results (crispy_asr_transcribe_recording: commands/transcription.rs:251,359,402)
int transcribe_batch_impl(crispy_asr* h, const float* const* pcm, const size_t* n, int batch, const crispy_asr_opts* opts,
                          crispy_asr_result** results, const volatile int* cancel) {
  if (!h || !results) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: NULL argument");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: batch < 0");
  for (int i = 0; i < batch; ++i) results[i] = nullptr;
  if (batch == 0) return CRISPY_OK;
  if (!pcm || !n) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: NULL argument");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_transcribe_batch: model not finalized");
  // empty clips produce empty results without touching the GPU (managers/transcription.rs:175-177); so do clips
  // shorter than 1 s = 100 mel frames, which whisper.cpp's whisper_full refuses ("input is too short", returns no
  // segments) [UPSTREAM-RECALL] -- the 168 samples the 48 -> 16 kHz resampler leaves past a 30 s chunk are such a clip
  std::vector<int> live;
  size_t stride = 1;
  for (int i = 0; i < batch; ++i) {
    if (n[i] > 480000)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: clip %d has %zu samples; the caller chunks at 480000 "
                  "(commands/transcription.rs:249-302)", i, n[i]);
    if (n[i] > 0) {
      if (!pcm[i]) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: clip %d is NULL", i);
      if (1 + ((long)n[i] + 200 - 400) / 160 < TS_DELTA_MIN) continue;
      live.push_back(i);
      if (n[i] > stride) stride = n[i];
    }
  }
  std::vector<crispy_asr_result_impl*> impl(batch, nullptr);
  auto cleanup = [&]() { for (auto* r : impl) delete r; for (int i = 0; i < batch; ++i) results[i] = nullptr; };
  for (int i = 0; i < batch; ++i) {
    impl[i] = new (std::nothrow) crispy_asr_result_impl();
    if (!impl[i]) { cleanup(); return fail(CRISPY_ERR_OOM, "crispy_asr_transcribe_batch: host allocation failed"); }
  }
  const int nb = (int)live.size();
  if (nb > 0) {
    const Special sp = special_tokens(h);
    const bool timestamps = !(opts && opts->no_timestamps) && sp.beg + 1501 <= h->hp.n_vocab;
    if (opts && (opts->beam_size < 0 || opts->beam_size > TS_MAX_CAND)) {
      cleanup();
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: beam_size %d; 0 .. %d (WHISPER_MAX_DECODERS)", opts->beam_size, TS_MAX_CAND);
    }
    if (opts && opts->beam_size > 1 && !timestamps) {
      cleanup();
      return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_transcribe_batch: beam search runs inside whisper_full's window loop (timestamps on)");
    }
    if (opts && (opts->n_initial_prompt < 0 || (opts->n_initial_prompt > 0 && !opts->initial_prompt))) {
      cleanup();
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: initial_prompt is NULL or its count negative");
    }
    if (opts && opts->carry_context && batch != 1) {
      cleanup();
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: carry_context needs a single-chunk call (batch %d)", batch);
    }
    if (opts)
      for (int i = 0; i < opts->n_initial_prompt; ++i)
        if (opts->initial_prompt[i] < 0 || opts->initial_prompt[i] >= h->hp.n_vocab) {
          cleanup();
          return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: initial prompt token %d out of range", opts->initial_prompt[i]);
        }
    std::vector<int> prompt = {sp.sot};
    if (sp.multilingual) {
      prompt.push_back(opts && opts->language_token > 0 ? opts->language_token : sp.lang0);   // <|en|>
      prompt.push_back(opts && opts->translate ? sp.translate : sp.transcribe);
    }
    if (!timestamps) prompt.push_back(sp.not_);
    int max_new = opts && opts->max_new_tokens > 0 ? opts->max_new_tokens
                                                   : (timestamps ? h->hp.n_text_ctx / 2 - 4 : h->hp.n_text_ctx / 2);
    if ((int)prompt.size() + max_new > h->hp.n_text_ctx) max_new = h->hp.n_text_ctx - (int)prompt.size();
    std::vector<int> lens(nb), lang(nb, 0);
    for (int k = 0; k < nb; ++k) lens[k] = (int)n[live[k]];
    const bool detect = sp.multilingual && !(opts && opts->language_token > 0);
    auto run = [&]() -> int {
      HIP_TRY(hipSetDevice(h->device));
      int rc = reserve_enc(h, nb);
      if (rc != CRISPY_OK) return rc;
      if (!h->w_pcm || (long)stride > h->cap_pcm_stride) {
        if (h->w_pcm) (void)hipFree(h->w_pcm);
        h->w_pcm = nullptr;
        HIP_TRY(hipMalloc(&h->w_pcm, (size_t)h->cap_batch * stride * sizeof(float)));
        h->cap_pcm_stride = (long)stride;
      }
      // every clip straight from the caller's memory into its row (no packed host copy: for the 21 chunks of a ten-minute
      // recording that was 40 MB zero-filled, copied and then copied again); what lies behind a clip's end in its row is
      // never read -- the log-mel takes n_samples per clip
      for (int k = 0; k < nb; ++k)
        HIP_TRY(hipMemcpyAsync(h->w_pcm + (size_t)k * stride, pcm[live[k]], (size_t)lens[k] * sizeof(float), hipMemcpyHostToDevice, h->stream));
      rc = crispy_mel_compute_device(h->mel, h->w_pcm, (long)stride, lens.data(), nb, nullptr, h->w_melt, h->stream);
      if (rc != CRISPY_OK) return rc;
      rc = crispy_asr_encode_device(h, h->w_melt, nb, h->w_enc, h->stream);
      if (rc != CRISPY_OK) return rc;
      if (detect) {
        rc = crispy_asr_detect_language_device(h, h->w_enc, nb, lang.data());
        if (rc != CRISPY_OK) return rc;
      } else if (sp.multilingual) {
        std::fill(lang.begin(), lang.end(), prompt[1]);
      }
      for (int k = 0; k < nb; ++k) impl[live[k]]->language_token = lang[k];
      if (!timestamps) {
        std::vector<int> toks((size_t)nb * max_new), n_out(nb, 0);
        rc = crispy_asr_decode_greedy_lang_device(h, h->w_enc, nb, prompt.data(), (int)prompt.size(),
                                                  detect ? lang.data() : nullptr, max_new, toks.data(), n_out.data(), nullptr);
        if (rc != CRISPY_OK) return rc;
        for (int k = 0; k < nb; ++k) {
          crispy_asr_result_impl* r = impl[live[k]];
          r->tokens.assign(toks.begin() + (size_t)k * max_new, toks.begin() + (size_t)k * max_new + n_out[k]);
          for (int t : r->tokens)
            if (t < h->eot && t < (int)h->vocab.size()) r->text += h->vocab[t];
        }
        return CRISPY_OK;
      }
      // ---- whisper_full's seek loop, all clips in lock step ----
      // [UPSTREAM-RECALL: whisper_full_with_state].  Per round every clip that has audio left decodes one window:
      //   * prompt = (<|startofprev|> + the last min(n_text_ctx / 2, |past|) tokens of the text so far) + the usual prompt;
      //     the past is dropped when fewer than 5 s of audio are left ("a very short segment ... tends to confuse the
      //     decoder") and for a re-decode at a temperature >= 0.5; after a window: past = the past part of its prompt + its
      //     kept tokens (nothing from a window dropped as silence);
      //   * the temperature ladder: greedy at `temperature`, all clips of the round as ONE batch (their prompts differ in
      //     length: decode_ts left-pads); a clip whose window fails is decoded again at the next temperature with
      //     best_of sampling decoders (rows of one batch over copies of its encoder output), until one passes or the
      //     ladder ends;
      //   * no-speech rule, segments, and how far the window advances (the last closed timestamp pair, the whole
      //     window after a single closing timestamp).
      const int delta_min = TS_DELTA_MIN;
      std::vector<int> seek(nb, 0), seek_end(nb);
      for (int k = 0; k < nb; ++k) seek_end[k] = 1 + (lens[k] + 200 - 400) / 160;      // whisper.cpp's mel.n_len_org
      const float t0 = opts ? opts->temperature : 0.f;
      const float t_inc = !opts || opts->temperature_inc == 0.f ? 0.2f : opts->temperature_inc;
      const float entropy_thold = !opts || opts->entropy_thold == 0.f ? 2.4f : opts->entropy_thold;
      const float logprob_thold = !opts || opts->logprob_thold == 0.f ? -1.0f : opts->logprob_thold;
      const float no_speech_thold = !opts || opts->no_speech_thold == 0.f ? 0.6f : opts->no_speech_thold;
      const int best_of = std::max(1, !opts || opts->best_of == 0 ? 5 : opts->best_of);
      if (best_of > 8) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: best_of %d > 8 (WHISPER_MAX_DECODERS)", best_of);
      std::vector<float> temps;
      if (t_inc > 0.f) for (float t = t0; t < 1.0f + 1e-6f; t += t_inc) temps.push_back(t);
      else temps.push_back(t0);
      if (temps.empty()) temps.push_back(t0);
      const bool use_past = !(opts && opts->no_prev_text);
      // The conditioning text a chunk starts with [UPSTREAM-RECALL: whisper_full_with_state, prompt_past]: nothing
      // (no_context = true, whisper.cpp's default); with carry_context what the previous call on this handle ended with;
      // the caller's initial prompt rotated in front of that.
      std::vector<std::vector<int>> past(nb);
      {
        std::vector<int> start;
        if (opts && opts->n_initial_prompt > 0) start.assign(opts->initial_prompt, opts->initial_prompt + opts->n_initial_prompt);
        if (opts && opts->carry_context) start.insert(start.end(), h->prompt_past.begin(), h->prompt_past.end());
        for (int k = 0; k < nb; ++k) past[k] = start;
      }
      const unsigned char *ts_mask = h->d_ts_mask, *ts_mask_first = h->d_ts_mask_first;
      if (opts && opts->suppress_nst) {
        rc = build_nst_masks(h);
        if (rc != CRISPY_OK) return rc;
        ts_mask = h->d_ts_mask_nst; ts_mask_first = h->d_ts_mask_first_nst;
      }
      // whisper.cpp's BEAM_SEARCH strategy (beam_size > 1): beam_size decoders at temperature 0, best_of above, every pass through
      // decode_beam (candidates drawn per decoder, sorted, dealt; see there); 0 / 1: the GREEDY strategy
      const int beam = opts && opts->beam_size > 1 ? opts->beam_size : 0;
      std::vector<std::vector<std::mt19937>> rngs(nb);
      for (int k = 0; k < nb; ++k)
        for (int j = 0; j < std::max(best_of, beam); ++j) rngs[k].emplace_back((unsigned)j);
      const int n_init = (int)prompt.size();
      const size_t enc_clip = (size_t)h->hp.n_audio_ctx * h->hp.n_audio_state;
      // rows of one fallback decode: whole clips x best_of
      const int kLadderRows = kLadderRowsMax;
      // the decoder workspace for the widest pass of this call, taken once: growing it between the greedy pass and the
      // first fallback pass freed every buffer and dropped the captured steps.  Widest = the most rows any pass
      // of the ladder can have: the beam pass at temperature 0 (groups of kLadderRows / beam clips x beam rows) and the
      // best_of passes above it (kLadderRows / best_of clips x best_of rows) -- ADVICE r5: with beam < best_of the beam
      // pass is the wider one.
      {
        auto pass_rows = [&](int n_dec) { return std::min(nb * n_dec, std::max(1, kLadderRows / n_dec) * n_dec); };
        int rows_max = nb;
        if (temps.size() > 1 && best_of > 1) rows_max = std::max(rows_max, pass_rows(best_of));
        if (beam > 1) rows_max = std::max(rows_max, pass_rows(beam));
        if (rows_max > nb) {
          rc = reserve_dec(h, rows_max, nb);
          if (rc != CRISPY_OK) return rc;
        }
      }
      // the encoder outputs of a group of fallback clips, gathered (one per clip).  The group size changes from pass to pass
      // (every pending clip in one group at n_dec == 1, kLadderRows / n_dec otherwise): the buffer is kept by capacity and
      // regrown -- round 5 sized it from the first group that needed it, and a later, larger group overflowed it
      float* d_enc_rep = nullptr;
      int enc_rep_clips = 0;
      struct RepGuard { float** p; ~RepGuard() { if (*p) (void)hipFree(*p); } } rep_guard{&d_enc_rep};
      auto reserve_enc_rep = [&](int n_clips) -> int {
        if (n_clips <= enc_rep_clips) return CRISPY_OK;
        if (d_enc_rep) {                                    // copies into / decodes from the old buffer may be in flight
          HIP_TRY(hipStreamSynchronize(h->stream));
          (void)hipFree(d_enc_rep);
          d_enc_rep = nullptr; enc_rep_clips = 0;
        }
        HIP_TRY(hipMalloc(&d_enc_rep, (size_t)n_clips * enc_clip * sizeof(float)));
        enc_rep_clips = n_clips;
        return CRISPY_OK;
      };
      auto build_prompt = [&](int k, int lang_tok, float t_cur) {
        std::vector<int> p;
        if (use_past && !past[k].empty() && t_cur < 0.5f) {
          int n_take = std::min<int>(h->hp.n_text_ctx / 2, (int)past[k].size());
          n_take = std::min(n_take, h->hp.n_text_ctx - max_new - n_init - 1);
          if (n_take > 0) {
            p.push_back(sp.prev);
            p.insert(p.end(), past[k].end() - n_take, past[k].end());
          }
        }
        p.insert(p.end(), prompt.begin(), prompt.end());
        if (sp.multilingual) p[p.size() - n_init + 1] = lang_tok;
        return p;
      };
      // whisper.cpp loops until seek + delta_min >= seek_end.  Every round advances every active clip by seek_delta >= 2
      // (a closed pair ends on a timestamp strictly above <|0.00|>, otherwise the delta is the whole window), so
      // 1500 rounds cover any 30 s clip; running out of them is reported, never a silently shorter transcript.
      const int kMaxRounds = 1501;
      for (int round = 0;; ++round) {
        if (round >= kMaxRounds)
          return fail(CRISPY_ERR_HIP, "crispy_asr_transcribe_batch: seek loop did not terminate after %d windows", kMaxRounds);
        if (cancel && *cancel) return fail(CRISPY_ERR_CANCELLED, "transcription cancelled by the caller");
        std::vector<int> act;
        for (int k = 0; k < nb; ++k)
          if (seek_end[k] >= delta_min && seek[k] + delta_min < seek_end[k]) act.push_back(k);   // < 100 ms left: whisper.cpp stops
        if (act.empty()) break;
        const int na = (int)act.size();
        if (!(round == 0 && na == nb)) {     // round 0 with every clip active: the encoder output is already there
          std::vector<int> sk(na);
          for (int a = 0; a < na; ++a) sk[a] = seek[act[a]];
          rc = crispy_mel_window_device(h->mel, act.data(), sk.data(), na, nullptr, h->w_melt, h->stream);
          if (rc != CRISPY_OK) return rc;
          rc = crispy_asr_encode_device(h, h->w_melt, na, h->w_enc, h->stream);
          if (rc != CRISPY_OK) return rc;
        }
        for (int a = 0; a < na; ++a) {
          const int k = act[a];
          if (seek[k] > 0 && seek[k] + 500 >= seek_end[k]) past[k].clear();
        }
        // per active clip: the pass whisper_full ends up accepting
        struct Accepted {
          std::vector<int> toks, tids, prompt;
          std::vector<float> plog;
          DecoderPass d;
          float nosp = 0.f, temperature = 0.f;
          int decoder = 0;
          bool have = false;
        };
        std::vector<Accepted> acc((size_t)na);
        std::vector<int> pending((size_t)na);
        for (int a = 0; a < na; ++a) pending[a] = a;
        for (size_t it = 0; it < temps.size() && !pending.empty(); ++it) {
          const float t_cur = temps[it];
          const bool last_temp = it + 1 == temps.size();
          const int n_dec = t_cur > 0.f ? best_of : (beam ? beam : 1);
          std::vector<int> still;
          // Groups of clips decoded together, n_dec rows each (rows [c n_dec, (c + 1) n_dec) of a group are the decoders of its
          // clip c: one cross K|V per clip, decode_ts's xgroup).  At temperature 0 that is every pending clip in one group, one
          // row each, straight off h->w_enc while nothing has dropped out; above it the pending clips x best_of, in groups of
          // at most kLadderRows rows -- ALL of them side by side, not one clip after the other (a batch in
          // which a third of the windows fall back used to decode them one by one, five rows at a time).
          const int per_group = n_dec == 1 && !beam ? (int)pending.size() : std::max(1, kLadderRows / n_dec);
          std::vector<std::vector<int>> groups;
          for (size_t g0 = 0; g0 < pending.size(); g0 += (size_t)per_group)
            groups.emplace_back(pending.begin() + g0, pending.begin() + std::min(pending.size(), g0 + (size_t)per_group));
          for (const std::vector<int>& grp : groups) {
            const int n_clips = (int)grp.size(), rows = n_clips * n_dec;
            const float* d_enc = h->w_enc;
            bool contiguous = true;               // the group's clips are w_enc's first n_clips, in order
            for (int c = 0; c < n_clips; ++c) contiguous = contiguous && grp[c] == c;
            if (!contiguous) {                    // one copy per CLIP (its decoders share it)
              rc = reserve_enc_rep(n_clips);
              if (rc != CRISPY_OK) return rc;
              for (int c = 0; c < n_clips; ++c)
                HIP_TRY(hipMemcpyAsync(d_enc_rep + (size_t)c * enc_clip, h->w_enc + (size_t)grp[c] * enc_clip, enc_clip * sizeof(float),
                                       hipMemcpyDeviceToDevice, h->stream));
              d_enc = d_enc_rep;
            }
            std::vector<std::vector<int>> prompts((size_t)rows);
            std::vector<int> r_seek(rows), r_end(rows);
            for (int r = 0; r < rows; ++r) {
              const int k = act[grp[r / n_dec]];
              prompts[r] = build_prompt(k, lang[k], t_cur);
              r_seek[r] = seek[k]; r_end[r] = seek_end[k];
            }
            std::vector<double> u;
            if (t_cur > 0.f && !beam) {   // the variates decoder j of clip k would draw, from a copy of ITS generator
              u.resize((size_t)max_new * rows);
              for (int r = 0; r < rows; ++r) {
                std::mt19937 g = rngs[act[grp[r / n_dec]]][r % n_dec];
                for (int i = 0; i < max_new; ++i) u[(size_t)i * rows + r] = canonical(g);
              }
            }
            std::vector<int> toks((size_t)rows * max_new), tids((size_t)rows * max_new), n_out(rows, 0);
            std::vector<float> plog((size_t)rows * max_new), nosp(rows, 0.f);
            if (beam) {
              std::vector<std::vector<int>> clip_prompts((size_t)n_clips);
              std::vector<int> c_seek(n_clips), c_end(n_clips);
              std::vector<std::mt19937*> row_rng((size_t)rows);
              for (int c = 0; c < n_clips; ++c) {
                const int k = act[grp[c]];
                clip_prompts[c] = prompts[(size_t)c * n_dec];
                c_seek[c] = seek[k]; c_end[c] = seek_end[k];
                for (int j = 0; j < n_dec; ++j) row_rng[(size_t)c * n_dec + j] = &rngs[k][j];
              }
              rc = decode_beam(h, d_enc, n_clips, n_dec, beam, clip_prompts, TS_RULES_WCPP, c_seek.data(), c_end.data(), max_new, ts_mask,
                               ts_mask_first, t_cur, row_rng, toks.data(), tids.data(), plog.data(), nosp.data(), n_out.data());
            } else {
              rc = decode_ts(h, d_enc, rows, prompts, TS_RULES_WCPP, r_seek.data(), r_end.data(), max_new, ts_mask,
                             ts_mask_first, t_cur, t_cur > 0.f ? u.data() : nullptr, toks.data(), tids.data(), plog.data(),
                             nosp.data(), n_out.data(), n_dec);
            }
            if (rc != CRISPY_OK) return rc;
            // evaluate: per clip of the group, its n_dec decoders
            for (int c = 0; c < n_clips; ++c) {
              const int a = grp[c], k = act[a];
              std::vector<DecoderPass> decs((size_t)n_dec);
              for (int j = 0; j < n_dec; ++j) {
                const int r = c * n_dec + j;
                DecoderPass& d = decs[j];
                d.toks = toks.data() + (size_t)r * max_new;
                d.tids = tids.data() + (size_t)r * max_new;
                d.plog = plog.data() + (size_t)r * max_new;
                d.n = n_out[r];
                replay_decoder(d, max_new, sp.beg, h->eot, seek[k], seek_end[k], delta_min);
                if (t_cur > 0.f && !beam) rngs[k][j].discard(2ull * (unsigned long long)d.n);      // what it drew: two per pick (a beam pass drew from the generators themselves)
              }
              // rank the sequences that did not fail (whisper.cpp: "rank the resulting sequences and select the best one")
              int best = acc[a].have ? acc[a].decoder : 0;      // best_decoder_id survives a pass in which every decoder failed
              if (best >= n_dec) best = 0;
              double best_score = -INFINITY;
              for (int j = 0; j < n_dec; ++j) {
                DecoderPass& d = decs[j];
                if (d.failed) continue;
                score_decoder(d);
                if (d.result_len > 32 && d.entropy < entropy_thold) { d.failed = true; continue; }
                if (best_score < d.score) { best_score = d.score; best = j; }
              }
              const DecoderPass& bd = decs[best];
              const float clip_nosp = nosp[c * n_dec];          // every decoder of a clip saw the same prompt logits
              bool success = true;
              if (!last_temp && (bd.failed || (bd.avg_logprobs < logprob_thold && clip_nosp < no_speech_thold)))
                success = false;
              Accepted& A = acc[a];
              const int r = c * n_dec + best;
              A.toks.assign(toks.begin() + (size_t)r * max_new, toks.begin() + (size_t)r * max_new + bd.n);
              A.tids.assign(tids.begin() + (size_t)r * max_new, tids.begin() + (size_t)r * max_new + bd.n);
              A.plog.assign(plog.begin() + (size_t)r * max_new, plog.begin() + (size_t)r * max_new + bd.n);
              A.d = bd;
              A.d.toks = A.toks.data(); A.d.tids = A.tids.data(); A.d.plog = A.plog.data();
              A.prompt = prompts[r];
              A.nosp = clip_nosp;
              A.temperature = t_cur;
              A.decoder = best;
              A.have = true;
              if (!success) still.push_back(a);
            }
          }
          pending.swap(still);
        }
        for (int a = 0; a < na; ++a) {
          const int k = act[a];
          crispy_asr_result_impl* r = impl[live[k]];
          const Accepted& A = acc[a];
          const DecoderPass& d = A.d;
          // a decoder that failed before the ranking keeps all its tokens (only ranked sequences are cut to result_len)
          const int n_cur = d.scored ? d.result_len : d.n;
          const bool is_no_speech = A.nosp > no_speech_thold && d.avg_logprobs < logprob_thold;
          {
            std::vector<int> np;
            if (A.prompt.front() == sp.prev) np.assign(A.prompt.begin() + 1, A.prompt.end() - n_init);
            if (!is_no_speech) np.insert(np.end(), A.toks.begin(), A.toks.begin() + d.result_len);
            past[k].swap(np);
          }
          int seek_delta = d.seek_delta;
          if (n_cur > 0 && !is_no_speech) {
            window_segments(h, A.toks.data(), A.tids.data(), n_cur, sp.beg, seek[k], seek_delta, r);
            for (int i = 0; i < n_cur; ++i)
              if (A.toks[i] != h->eot) r->tokens.push_back(A.toks[i]);
          }
          // a single closing timestamp: nothing is left to say in this chunk [UPSTREAM-RECALL: whisper.cpp PR 2629]
          if (n_cur > 1 && A.toks[n_cur - 2] < sp.beg && A.toks[n_cur - 1] > sp.beg)
            seek_delta = std::min(seek_end[k] - seek[k], 3000);
          crispy_asr_window w{};
          w.seek = seek[k]; w.seek_advance = seek_delta; w.n_tokens = is_no_speech ? 0 : n_cur; w.decoder = A.decoder;
          w.failed = d.failed ? 1 : 0; w.no_speech = is_no_speech ? 1 : 0; w.temperature = A.temperature;
          w.no_speech_prob = A.nosp; w.avg_logprob = (float)d.avg_logprobs; w.entropy = (float)d.entropy;
          r->wins.push_back(w);
          seek[k] += seek_delta;
        }
      }
      for (int k = 0; k < nb; ++k) {
        crispy_asr_result_impl* r = impl[live[k]];
        for (const std::string& t : r->seg_text) r->text += t;
      }
      if (batch == 1) h->prompt_past = past[0];      // whisper.cpp keeps prompt_past in the state; the next call uses it only with carry_context
      return CRISPY_OK;
    };
    const int rc = run();
    if (rc != CRISPY_OK) { cleanup(); return rc; }
  }
  for (int i = 0; i < batch; ++i) {
    publish(impl[i]);
    results[i] = &impl[i]->pub;
  }
  return CRISPY_OK;
}

// Rust's str::trim(): the code points with the White_Space property, off both ends of a UTF-8 string
// (managers/transcription.rs:187 trims every chunk's text; commands/transcription.rs:276 tests `trim().is_empty()`)
bool unicode_space(unsigned cp) {
  return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
         cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
std::string trim_unicode(const std::string& s) {
  auto decode = [&](size_t i, size_t* len) -> unsigned {      // one code point at byte i (malformed bytes stand for themselves)
    const unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
    if (c < 0x80) { *len = 1; return c; }
    if ((c & 0xE0) == 0xC0 && cont(1)) { *len = 2; return ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); }
    if ((c & 0xF0) == 0xE0 && cont(1) && cont(2)) {
      *len = 3;
      return ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
    }
    *len = 1;
    return 0xFFFFFFFFu;
  };
  size_t a = 0, b = s.size();
  while (a < b) {
    size_t len = 1;
    if (!unicode_space(decode(a, &len))) break;
    a += len;
  }
  while (b > a) {
    size_t k = b - 1;
    while (k > a && ((unsigned char)s[k] & 0xC0) == 0x80 && b - k < 3) --k;      // back to the lead byte of the last code point
    size_t len = 1;
    const unsigned cp = decode(k, &len);
    if (k + len != b || !unicode_space(cp)) break;
    b = k;
  }
  return s.substr(a, b - a);
}

}  // namespace

extern "C" {

// `run_transcription`'s chunk loop (commands/transcription.rs:249-302, 363-400, 468) over a whole 16 kHz recording, with
// the chunks decoded TOGETHER: the reference's loop is serial because its engine is (one chunk per call under a mutex,
// managers/transcription.rs:27,178), yet the chunks are independent -- TranscribeOptions::default() carries no context
// from chunk to chunk (:184) -- so a group of them is one batch call and an hour of audio is one or two calls, not 120.
int crispy_asr_transcribe_recording(crispy_asr* h, const float* pcm16k, size_t n, const crispy_asr_opts* opts, int max_batch,
                                    const volatile int* cancel_flag, crispy_asr_progress_fn progress, void* progress_user,
                                    crispy_asr_result** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_recording: out is NULL");
  *out = nullptr;
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_recording: NULL handle");
  if (n > 0 && !pcm16k) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_recording: NULL audio");
  if (max_batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_recording: max_batch < 0");
  if (opts && opts->carry_context)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_recording: carry_context is a single-chunk option (the chunks of a "
                "recording are decoded side by side, each from a clean context, as the reference does)");
  if (!h->finalized) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_transcribe_recording: model not finalized");
  constexpr size_t kChunk = 480000;                        // 30 s at 16 kHz (commands/transcription.rs:175-176)
  const size_t n_chunks = (n + kChunk - 1) / kChunk;       // the last partial chunk is passed as it is (the engine pads)
  const size_t group = max_batch > 0 ? (size_t)max_batch : 128;
  crispy_asr_result_impl* R = new (std::nothrow) crispy_asr_result_impl();
  if (!R) return fail(CRISPY_ERR_OOM, "crispy_asr_transcribe_recording: host allocation failed");
  struct Own { crispy_asr_result_impl* r; ~Own() { delete r; } } own{R};
  bool first_text = true;
  for (size_t g0 = 0; g0 < n_chunks; g0 += group) {
    if (cancel_flag && *cancel_flag) return fail(CRISPY_ERR_CANCELLED, "transcription cancelled by the caller");
    const int nb = (int)std::min(group, n_chunks - g0);
    std::vector<const float*> ptrs((size_t)nb);
    std::vector<size_t> lens((size_t)nb);
    std::vector<crispy_asr_result*> res((size_t)nb, nullptr);
    for (int i = 0; i < nb; ++i) {
      const size_t at = (g0 + (size_t)i) * kChunk;
      ptrs[i] = pcm16k + at;
      lens[i] = std::min(kChunk, n - at);
    }
    const int rc = transcribe_batch_impl(h, ptrs.data(), lens.data(), nb, opts, res.data(), cancel_flag);
    if (rc != CRISPY_OK) return rc;
    for (int i = 0; i < nb; ++i) {
      const crispy_asr_result& r = *res[i];
      const size_t ci = g0 + (size_t)i;
      const std::string t = trim_unicode(r.text ? r.text : "");
      if (!t.empty()) {                                    // parts.push(..) only for non-blank chunk texts; joined with " "
        if (!first_text) R->text += ' ';
        R->text += t;
        first_text = false;
      }
      R->tokens.insert(R->tokens.end(), r.tokens, r.tokens + r.n_tokens);
      if (ci == 0) R->language_token = r.language_token;
      const float t_off = (float)(ci * 30.0);              // chunk_start_seconds (transcription.rs:262)
      for (int k = 0; k < r.n_segments; ++k) {
        R->seg_t0.push_back(t_off + r.segments[k].t0);
        R->seg_t1.push_back(t_off + r.segments[k].t1);
        R->seg_text.emplace_back(r.segments[k].text ? r.segments[k].text : "");
      }
      for (int k = 0; k < r.n_windows; ++k) {
        crispy_asr_window w = r.windows[k];
        w.seek += (int)(ci * 3000);
        R->wins.push_back(w);
      }
      crispy_asr_free_result(res[i]);
    }
    if (progress) progress(std::min(n, (g0 + (size_t)nb) * kChunk), n, progress_user);
  }
  publish(R);
  own.r = nullptr;
  *out = &R->pub;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_transcribe_recording")

void crispy_asr_free_result(crispy_asr_result* r) try {
  if (!r) return;
  delete reinterpret_cast<crispy_asr_result_impl*>(r);   // pub is the first member
} CRISPY_CATCH_VOID("crispy_asr_free_result")

}  // extern "C"
