// whisper_full.cpp -- whisper.cpp's whisper_full_with_state on the GPU path [UPSTREAM-RECALL]: the seek loop over 30 s windows,
// previous-text conditioning, the temperature ladder with best_of sampling decoders (or beam search), the no-speech rule,
// segments; all clips of a batch in lock step.  `crispy_asr_transcribe{,_batch}` = engine.transcribe(&audio,
// &TranscribeOptions::default()) (src-tauri/src/managers/transcription.rs:183-185); `crispy_asr_transcribe_recording` = the chunk
// loop of run_transcription (src-tauri/src/commands/transcription.rs:249-302, 363-400, 468).
#include "whisper_internal.h"

using namespace crispy;
using namespace crispy::asr;

namespace crispy {
namespace asr {
namespace {

int transcribe_batch_impl(crispy_asr* h, const float* const* pcm, const size_t* n, int batch, const crispy_asr_opts* opts,
                          crispy_asr_result** results, const volatile int* cancel);

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

}  // namespace

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

namespace {

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
}  // namespace asr
}  // namespace crispy

extern "C" {

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

namespace crispy {
namespace asr {
namespace {

// ---- engine.transcribe for a batch of chunks: one object per call, one short method per step of whisper_full ----
// (Round 6: this was one 400-line function.  The statements are the same; what each group of them is FOR is now its name.)

// the pass whisper_full ends up accepting for one active clip of a round
struct Accepted {
  std::vector<int> toks, tids, prompt;
  std::vector<float> plog;
  DecoderPass d;
  float nosp = 0.f, temperature = 0.f;
  int decoder = 0;
  bool have = false;
};

// what one decode of a group of clips (n_dec rows each) returned
struct GroupPicks {
  int max_new = 0;
  std::vector<int> toks, tids, n_out;
  std::vector<float> plog, nosp;
  std::vector<std::vector<int>> prompts;      // per row
};

class BatchCall {
 public:
  BatchCall(crispy_asr* h_, const float* const* pcm_, const size_t* n_, int batch_, const crispy_asr_opts* opts_, const volatile int* cancel_)
      : h(h_), pcm(pcm_), n(n_), batch(batch_), opts(opts_), cancel(cancel_), sp(special_tokens(h_)) {}
  ~BatchCall() {
    for (auto* r : impl) delete r;
    if (d_enc_rep) (void)hipFree(d_enc_rep);
  }
  int prepare();                       // argument checks, which clips are live, the prompt, one result object per clip
  int run();                           // everything on the device
  void release(crispy_asr_result** results) {      // hands the results over (the destructor then has nothing to delete)
    for (int i = 0; i < batch; ++i) {
      publish(impl[i]);
      results[i] = &impl[i]->pub;
      impl[i] = nullptr;
    }
  }

 private:
  int check_options() const;
  int upload_and_encode();
  int decode_plain();                  // no_timestamps = 1: one window, plain greedy arg-max
  int setup_ladder();                  // thresholds, temperatures, masks, generators, the widest pass's workspace
  int seek_loop();
  int decode_round(const std::vector<int>& act, std::vector<Accepted>& acc);
  int decode_group(const std::vector<int>& act, const std::vector<int>& grp, float t_cur, int n_dec, GroupPicks& out);
  bool evaluate_clip(int a, int k, int c, float t_cur, bool last_temp, int n_dec, const GroupPicks& g, Accepted& A);
  void finish_window(int k, const Accepted& A);
  std::vector<int> build_prompt(int k, int lang_tok, float t_cur) const;
  int reserve_enc_rep(int n_clips);

  crispy_asr* h;
  const float* const* pcm;
  const size_t* n;
  int batch;
  const crispy_asr_opts* opts;
  const volatile int* cancel;
  const Special sp;
  std::vector<crispy_asr_result_impl*> impl;
  std::vector<int> live, lens, lang, prompt;
  size_t stride = 1;
  int nb = 0, max_new = 0, n_init = 0;
  bool timestamps = true, detect = false;
  // whisper_full's parameters (0 in crispy_asr_opts = whisper.cpp's default)
  float entropy_thold = 2.4f, logprob_thold = -1.0f, no_speech_thold = 0.6f;
  int best_of = 5, beam = 0;
  bool use_past = true;
  std::vector<float> temps;
  const unsigned char *ts_mask = nullptr, *ts_mask_first = nullptr;
  // per live clip
  std::vector<int> seek, seek_end;
  std::vector<std::vector<int>> past;
  std::vector<std::vector<std::mt19937>> rngs;
  // the encoder outputs of a group of fallback clips, gathered (one per clip).  The group size changes from pass to pass
  // (every pending clip in one group at n_dec == 1, kLadderRowsMax / n_dec otherwise): kept by capacity and regrown -- round 5
  // sized it from the first group that needed it, and a later, larger group overflowed it
  float* d_enc_rep = nullptr;
  int enc_rep_clips = 0;
  size_t enc_clip = 0;
};

int BatchCall::check_options() const {
  if (!opts) return CRISPY_OK;
  if (opts->beam_size < 0 || opts->beam_size > TS_MAX_CAND)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: beam_size %d; 0 .. %d (WHISPER_MAX_DECODERS)", opts->beam_size, TS_MAX_CAND);
  if (opts->beam_size > 1 && !timestamps)
    return fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_transcribe_batch: beam search runs inside whisper_full's window loop (timestamps on)");
  if (opts->n_initial_prompt < 0 || (opts->n_initial_prompt > 0 && !opts->initial_prompt))
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: initial_prompt is NULL or its count negative");
  if (opts->carry_context && batch != 1)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: carry_context needs a single-chunk call (batch %d)", batch);
  for (int i = 0; i < opts->n_initial_prompt; ++i)
    if (opts->initial_prompt[i] < 0 || opts->initial_prompt[i] >= h->hp.n_vocab)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: initial prompt token %d out of range", opts->initial_prompt[i]);
  return CRISPY_OK;
}

int BatchCall::prepare() {
  // empty clips produce empty results without touching the GPU (managers/transcription.rs:175-177); so do clips
  // shorter than 1 s = 100 mel frames, which whisper.cpp's whisper_full refuses ("input is too short", returns no
  // segments) [UPSTREAM-RECALL] -- the 168 samples the 48 -> 16 kHz resampler leaves past a 30 s chunk are such a clip
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
  impl.assign((size_t)batch, nullptr);
  for (int i = 0; i < batch; ++i) {
    impl[i] = new (std::nothrow) crispy_asr_result_impl();
    if (!impl[i]) return fail(CRISPY_ERR_OOM, "crispy_asr_transcribe_batch: host allocation failed");
  }
  nb = (int)live.size();
  if (nb == 0) return CRISPY_OK;
  timestamps = !(opts && opts->no_timestamps) && sp.beg + 1501 <= h->hp.n_vocab;
  const int rc = check_options();
  if (rc != CRISPY_OK) return rc;
  prompt = {sp.sot};
  if (sp.multilingual) {
    prompt.push_back(opts && opts->language_token > 0 ? opts->language_token : sp.lang0);   // <|en|>
    prompt.push_back(opts && opts->translate ? sp.translate : sp.transcribe);
  }
  if (!timestamps) prompt.push_back(sp.not_);
  n_init = (int)prompt.size();
  max_new = opts && opts->max_new_tokens > 0 ? opts->max_new_tokens : (timestamps ? h->hp.n_text_ctx / 2 - 4 : h->hp.n_text_ctx / 2);
  if (n_init + max_new > h->hp.n_text_ctx) max_new = h->hp.n_text_ctx - n_init;
  lens.resize(nb);
  lang.assign(nb, 0);
  for (int k = 0; k < nb; ++k) lens[k] = (int)n[live[k]];
  detect = sp.multilingual && !(opts && opts->language_token > 0);
  enc_clip = (size_t)h->hp.n_audio_ctx * h->hp.n_audio_state;
  return CRISPY_OK;
}

int BatchCall::upload_and_encode() {
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
  return CRISPY_OK;
}

int BatchCall::decode_plain() {
  std::vector<int> toks((size_t)nb * max_new), n_out(nb, 0);
  const int rc = crispy_asr_decode_greedy_lang_device(h, h->w_enc, nb, prompt.data(), n_init, detect ? lang.data() : nullptr, max_new,
                                                      toks.data(), n_out.data(), nullptr);
  if (rc != CRISPY_OK) return rc;
  for (int k = 0; k < nb; ++k) {
    crispy_asr_result_impl* r = impl[live[k]];
    r->tokens.assign(toks.begin() + (size_t)k * max_new, toks.begin() + (size_t)k * max_new + n_out[k]);
    for (int t : r->tokens)
      if (t < h->eot && t < (int)h->vocab.size()) r->text += h->vocab[t];
  }
  return CRISPY_OK;
}

int BatchCall::run() {
  if (nb == 0) return CRISPY_OK;
  int rc = upload_and_encode();
  if (rc != CRISPY_OK) return rc;
  if (!timestamps) return decode_plain();
  rc = setup_ladder();
  if (rc != CRISPY_OK) return rc;
  rc = seek_loop();
  if (rc != CRISPY_OK) return rc;
  for (int k = 0; k < nb; ++k) {
    crispy_asr_result_impl* r = impl[live[k]];
    for (const std::string& t : r->seg_text) r->text += t;
  }
  if (batch == 1) h->prompt_past = past[0];      // whisper.cpp keeps prompt_past in the state; the next call uses it only with carry_context
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
int BatchCall::setup_ladder() {
  seek.assign(nb, 0);
  seek_end.resize(nb);
  for (int k = 0; k < nb; ++k) seek_end[k] = 1 + (lens[k] + 200 - 400) / 160;      // whisper.cpp's mel.n_len_org
  const float t0 = opts ? opts->temperature : 0.f;
  const float t_inc = !opts || opts->temperature_inc == 0.f ? 0.2f : opts->temperature_inc;
  entropy_thold = !opts || opts->entropy_thold == 0.f ? 2.4f : opts->entropy_thold;
  logprob_thold = !opts || opts->logprob_thold == 0.f ? -1.0f : opts->logprob_thold;
  no_speech_thold = !opts || opts->no_speech_thold == 0.f ? 0.6f : opts->no_speech_thold;
  best_of = std::max(1, !opts || opts->best_of == 0 ? 5 : opts->best_of);
  if (best_of > 8) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_transcribe_batch: best_of %d > 8 (WHISPER_MAX_DECODERS)", best_of);
  if (t_inc > 0.f) for (float t = t0; t < 1.0f + 1e-6f; t += t_inc) temps.push_back(t);
  else temps.push_back(t0);
  if (temps.empty()) temps.push_back(t0);
  use_past = !(opts && opts->no_prev_text);
  // The conditioning text a chunk starts with [UPSTREAM-RECALL: whisper_full_with_state, prompt_past]: nothing
  // (no_context = true, whisper.cpp's default); with carry_context what the previous call on this handle ended with;
  // the caller's initial prompt rotated in front of that.
  past.assign(nb, {});
  {
    std::vector<int> start;
    if (opts && opts->n_initial_prompt > 0) start.assign(opts->initial_prompt, opts->initial_prompt + opts->n_initial_prompt);
    if (opts && opts->carry_context) start.insert(start.end(), h->prompt_past.begin(), h->prompt_past.end());
    for (int k = 0; k < nb; ++k) past[k] = start;
  }
  ts_mask = h->d_ts_mask; ts_mask_first = h->d_ts_mask_first;
  if (opts && opts->suppress_nst) {
    const int rc = build_nst_masks(h);
    if (rc != CRISPY_OK) return rc;
    ts_mask = h->d_ts_mask_nst; ts_mask_first = h->d_ts_mask_first_nst;
  }
  // whisper.cpp's BEAM_SEARCH strategy (beam_size > 1): beam_size decoders at temperature 0, best_of above, every pass through
  // decode_beam (candidates drawn per decoder, sorted, dealt; see there); 0 / 1: the GREEDY strategy
  beam = opts && opts->beam_size > 1 ? opts->beam_size : 0;
  rngs.assign(nb, {});
  for (int k = 0; k < nb; ++k)
    for (int j = 0; j < std::max(best_of, beam); ++j) rngs[k].emplace_back((unsigned)j);
  // the decoder workspace for the widest pass of this call, taken once: growing it between the greedy pass and the
  // first fallback pass freed every buffer and dropped the captured steps.  Widest = the most rows any pass
  // of the ladder can have: the beam pass at temperature 0 (groups of kLadderRowsMax / beam clips x beam rows) and the
  // best_of passes above it (kLadderRowsMax / best_of clips x best_of rows) -- ADVICE r5: with beam < best_of the beam
  // pass is the wider one.
  auto pass_rows = [&](int n_dec) { return std::min(nb * n_dec, std::max(1, kLadderRowsMax / n_dec) * n_dec); };
  int rows_max = nb;
  if (temps.size() > 1 && best_of > 1) rows_max = std::max(rows_max, pass_rows(best_of));
  if (beam > 1) rows_max = std::max(rows_max, pass_rows(beam));
  if (rows_max > nb) return reserve_dec(h, rows_max, nb);
  return CRISPY_OK;
}

int BatchCall::reserve_enc_rep(int n_clips) {
  if (n_clips <= enc_rep_clips) return CRISPY_OK;
  if (d_enc_rep) {                                    // copies into / decodes from the old buffer may be in flight
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(d_enc_rep);
    d_enc_rep = nullptr; enc_rep_clips = 0;
  }
  HIP_TRY(hipMalloc(&d_enc_rep, (size_t)n_clips * enc_clip * sizeof(float)));
  enc_rep_clips = n_clips;
  return CRISPY_OK;
}

std::vector<int> BatchCall::build_prompt(int k, int lang_tok, float t_cur) const {
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
}

int BatchCall::seek_loop() {
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
      if (seek_end[k] >= TS_DELTA_MIN && seek[k] + TS_DELTA_MIN < seek_end[k]) act.push_back(k);   // < 100 ms left: whisper.cpp stops
    if (act.empty()) return CRISPY_OK;
    const int na = (int)act.size();
    if (!(round == 0 && na == nb)) {     // round 0 with every clip active: the encoder output is already there
      std::vector<int> sk(na);
      for (int a = 0; a < na; ++a) sk[a] = seek[act[a]];
      int rc = crispy_mel_window_device(h->mel, act.data(), sk.data(), na, nullptr, h->w_melt, h->stream);
      if (rc != CRISPY_OK) return rc;
      rc = crispy_asr_encode_device(h, h->w_melt, na, h->w_enc, h->stream);
      if (rc != CRISPY_OK) return rc;
    }
    for (int a = 0; a < na; ++a) {
      const int k = act[a];
      if (seek[k] > 0 && seek[k] + 500 >= seek_end[k]) past[k].clear();
    }
    std::vector<Accepted> acc((size_t)na);
    const int rc = decode_round(act, acc);
    if (rc != CRISPY_OK) return rc;
    for (int a = 0; a < na; ++a) finish_window(act[a], acc[a]);
  }
}

// the temperature ladder of one round: per active clip, the pass whisper_full ends up accepting
int BatchCall::decode_round(const std::vector<int>& act, std::vector<Accepted>& acc) {
  const int na = (int)act.size();
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
    // at most kLadderRowsMax rows -- ALL of them side by side, not one clip after the other (a batch in
    // which a third of the windows fall back used to decode them one by one, five rows at a time).
    const int per_group = n_dec == 1 && !beam ? (int)pending.size() : std::max(1, kLadderRowsMax / n_dec);
    for (size_t g0 = 0; g0 < pending.size(); g0 += (size_t)per_group) {
      const std::vector<int> grp(pending.begin() + g0, pending.begin() + std::min(pending.size(), g0 + (size_t)per_group));
      GroupPicks picks;
      const int rc = decode_group(act, grp, t_cur, n_dec, picks);
      if (rc != CRISPY_OK) return rc;
      for (int c = 0; c < (int)grp.size(); ++c) {
        const int a = grp[c];
        if (!evaluate_clip(a, act[a], c, t_cur, last_temp, n_dec, picks, acc[a])) still.push_back(a);
      }
    }
    pending.swap(still);
  }
  return CRISPY_OK;
}

// one decode of the clips act[grp[c]], n_dec rows each, at t_cur
int BatchCall::decode_group(const std::vector<int>& act, const std::vector<int>& grp, float t_cur, int n_dec, GroupPicks& out) {
  const int n_clips = (int)grp.size(), rows = n_clips * n_dec;
  const float* d_enc = h->w_enc;
  bool contiguous = true;               // the group's clips are w_enc's first n_clips, in order
  for (int c = 0; c < n_clips; ++c) contiguous = contiguous && grp[c] == c;
  int rc;
  if (!contiguous) {                    // one copy per CLIP (its decoders share it)
    rc = reserve_enc_rep(n_clips);
    if (rc != CRISPY_OK) return rc;
    for (int c = 0; c < n_clips; ++c)
      HIP_TRY(hipMemcpyAsync(d_enc_rep + (size_t)c * enc_clip, h->w_enc + (size_t)grp[c] * enc_clip, enc_clip * sizeof(float),
                             hipMemcpyDeviceToDevice, h->stream));
    d_enc = d_enc_rep;
  }
  out.max_new = max_new;
  out.prompts.assign((size_t)rows, {});
  std::vector<int> r_seek(rows), r_end(rows);
  for (int r = 0; r < rows; ++r) {
    const int k = act[grp[r / n_dec]];
    out.prompts[r] = build_prompt(k, lang[k], t_cur);
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
  out.toks.assign((size_t)rows * max_new, 0); out.tids.assign((size_t)rows * max_new, 0); out.n_out.assign(rows, 0);
  out.plog.assign((size_t)rows * max_new, 0.f); out.nosp.assign(rows, 0.f);
  if (beam) {
    std::vector<std::vector<int>> clip_prompts((size_t)n_clips);
    std::vector<int> c_seek(n_clips), c_end(n_clips);
    std::vector<std::mt19937*> row_rng((size_t)rows);
    for (int c = 0; c < n_clips; ++c) {
      const int k = act[grp[c]];
      clip_prompts[c] = out.prompts[(size_t)c * n_dec];
      c_seek[c] = seek[k]; c_end[c] = seek_end[k];
      for (int j = 0; j < n_dec; ++j) row_rng[(size_t)c * n_dec + j] = &rngs[k][j];
    }
    return decode_beam(h, d_enc, n_clips, n_dec, beam, clip_prompts, TS_RULES_WCPP, c_seek.data(), c_end.data(), max_new, ts_mask,
                       ts_mask_first, t_cur, row_rng, out.toks.data(), out.tids.data(), out.plog.data(), out.nosp.data(), out.n_out.data());
  }
  return decode_ts(h, d_enc, rows, out.prompts, TS_RULES_WCPP, r_seek.data(), r_end.data(), max_new, ts_mask, ts_mask_first, t_cur,
                   t_cur > 0.f ? u.data() : nullptr, out.toks.data(), out.tids.data(), out.plog.data(), out.nosp.data(), out.n_out.data(), n_dec);
}

// clip k (active index a, clip c of the group): replay its n_dec decoders, rank them, record the best as the accepted pass;
// returns whether whisper_full is satisfied with it at this temperature
bool BatchCall::evaluate_clip(int a, int k, int c, float t_cur, bool last_temp, int n_dec, const GroupPicks& g, Accepted& A) {
  (void)a;
  const int mn = g.max_new;
  std::vector<DecoderPass> decs((size_t)n_dec);
  for (int j = 0; j < n_dec; ++j) {
    const int r = c * n_dec + j;
    DecoderPass& d = decs[j];
    d.toks = g.toks.data() + (size_t)r * mn;
    d.tids = g.tids.data() + (size_t)r * mn;
    d.plog = g.plog.data() + (size_t)r * mn;
    d.n = g.n_out[r];
    replay_decoder(d, mn, sp.beg, h->eot, seek[k], seek_end[k], TS_DELTA_MIN);
    if (t_cur > 0.f && !beam) rngs[k][j].discard(2ull * (unsigned long long)d.n);      // what it drew: two per pick (a beam pass drew from the generators themselves)
  }
  // rank the sequences that did not fail (whisper.cpp: "rank the resulting sequences and select the best one")
  int best = A.have ? A.decoder : 0;      // best_decoder_id survives a pass in which every decoder failed
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
  const float clip_nosp = g.nosp[c * n_dec];          // every decoder of a clip saw the same prompt logits
  const bool success = last_temp || !(bd.failed || (bd.avg_logprobs < logprob_thold && clip_nosp < no_speech_thold));
  const int r = c * n_dec + best;
  A.toks.assign(g.toks.begin() + (size_t)r * mn, g.toks.begin() + (size_t)r * mn + bd.n);
  A.tids.assign(g.tids.begin() + (size_t)r * mn, g.tids.begin() + (size_t)r * mn + bd.n);
  A.plog.assign(g.plog.begin() + (size_t)r * mn, g.plog.begin() + (size_t)r * mn + bd.n);
  A.d = bd;
  A.d.toks = A.toks.data(); A.d.tids = A.tids.data(); A.d.plog = A.plog.data();
  A.prompt = g.prompts[r];
  A.nosp = clip_nosp;
  A.temperature = t_cur;
  A.decoder = best;
  A.have = true;
  return success;
}

// the accepted pass of clip k's window into its result: conditioning text, segments, tokens, the window record, the seek advance
void BatchCall::finish_window(int k, const Accepted& A) {
  crispy_asr_result_impl* r = impl[live[k]];
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
  // At most kOneAnswerClips clips decode in lock step: that is the widest step the decode kernels take in the forms whose
  // arithmetic per row is the row's alone (SKINNY_MAX_M rows); a wider batch would put its greedy pass on other kernels and a
  // clip's last bits -- at a near tie its text -- would depend on whether 500 or 600 chunks were handed over.  More clips
  // are taken in turns; the results are released together or not at all.
  constexpr int kOneAnswerClips = SKINNY_MAX_M;
  for (int b0 = 0; b0 < batch; b0 += kOneAnswerClips) {
    const int nb = std::min(kOneAnswerClips, batch - b0);
    BatchCall call(h, pcm + b0, n + b0, nb, opts, cancel);
    int rc = call.prepare();
    if (rc == CRISPY_OK) rc = call.run();
    if (rc != CRISPY_OK) {                   // (the call object owns every result it made; the earlier turns' are given back)
      for (int i = 0; i < b0; ++i) { crispy_asr_free_result(results[i]); results[i] = nullptr; }
      return rc;
    }
    call.release(results + b0);
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
}  // namespace asr
}  // namespace crispy

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
  const size_t group = max_batch > 0 ? (size_t)max_batch : 128;       // (more than SKINNY_MAX_M are taken in turns by the batch call)
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
