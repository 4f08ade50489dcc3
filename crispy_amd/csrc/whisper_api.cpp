// whisper_api.cpp -- the Whisper model container behind include/crispy_hip.h: tensors by their model-file names and the copies
// derived from them (fused rows, LayerNorm folds, f16 operands of precision mode 1, packed operands of the fused decode step),
// the precision modes, the encoder, the workspaces.  Replaces transcribe_rs::whisper_cpp::WhisperEngine::{load, transcribe}
// together with ggml_load.cpp, decode_steps.cpp and whisper_full.cpp (reference: src-tauri/src/managers/transcription.rs:138-141,
// 183-185).
#include "whisper_internal.h"

using namespace crispy;
using namespace crispy::asr;

namespace crispy {
namespace asr {
namespace {

void add_spec(std::map<std::string, size_t>& spec, const std::string& name, size_t n) { spec[name] = n; }

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
}  // namespace

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
  if (h->d_beam_row) { (void)hipFree(h->d_beam_row); h->d_beam_row = nullptr; }
  if (h->d_beam_cand) { (void)hipFree(h->d_beam_cand); h->d_beam_cand = nullptr; }
  if (h->d_beam_rec_parent) { (void)hipFree(h->d_beam_rec_parent); h->d_beam_rec_parent = nullptr; }
  if (h->d_beam_u) { (void)hipFree(h->d_beam_u); h->d_beam_u = nullptr; h->beam_u_bytes = 0; }
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

}  // namespace asr
}  // namespace crispy

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

}  // extern "C"

namespace crispy {
namespace asr {
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
}  // namespace asr
}  // namespace crispy

extern "C" {

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

}  // extern "C"

namespace crispy {
namespace asr {
namespace {

// the convolution stem: padded frame-major mel -> residual stream h->w_x [batch * 1500][d] (GELU, positional embedding added)
int encode_stem(crispy_asr* h, const float* d_mel_t, int batch, hipStream_t s) {
  const int d = h->hp.n_audio_state, Tn = h->hp.n_audio_ctx, nm = h->hp.n_mels;
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
  return CRISPY_OK;
}

// the encoder layers over h->w_x in precision mode 1
int encode_layers_f16(crispy_asr* h, int batch, hipStream_t s) {
  const int d = h->hp.n_audio_state, Tn = h->hp.n_audio_ctx, H = h->hp.n_audio_head;
  const long rows = (long)batch * Tn;
  int rc = CRISPY_OK;
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
  return CRISPY_OK;
}

// ... and in precision mode 0 (f32 operands)
int encode_layers_f32(crispy_asr* h, int batch, hipStream_t s) {
  const int d = h->hp.n_audio_state, Tn = h->hp.n_audio_ctx, H = h->hp.n_audio_head;
  const long rows = (long)batch * Tn;
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
  return CRISPY_OK;
}

}  // namespace
}  // namespace asr
}  // namespace crispy

extern "C" {

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
  rc = encode_stem(h, d_mel_t, batch, s);
  if (rc == CRISPY_OK) rc = h->enc_precision == 1 ? encode_layers_f16(h, batch, s) : encode_layers_f32(h, batch, s);
  if (rc != CRISPY_OK) return rc;
  const long rows = (long)batch * h->hp.n_audio_ctx;
  const int d = h->hp.n_audio_state;
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
