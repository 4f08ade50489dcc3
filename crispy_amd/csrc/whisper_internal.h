// whisper_internal.h -- what the four translation units of the Whisper engine share (not part of the ABI):
//   whisper_api.cpp    the model container: tensors, derived copies, precision modes, encoder, workspaces
//   ggml_load.cpp      whisper.cpp GGML model files: reader, de-quantiser, resident block bookkeeping
//   decode_steps.cpp   the decoder: workspaces, the step forms (fused / matrix-vector / skinny), prompts, captured steps,
//                      the greedy / timestamp-rule / beam passes over one window
//   whisper_full.cpp   whisper_full on top of them: seek loop, temperature ladder, segments, results, the recording chunker
// Reference surface: transcribe_rs::whisper_cpp::WhisperEngine::{load, transcribe} (src-tauri/src/managers/transcription.rs:138-141,
// 183-185).
#pragma once
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

namespace crispy {
namespace asr {

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


}  // namespace asr
}  // namespace crispy

struct crispy_asr {
  int device = 0;
  crispy_asr_hparams hp{};
  hipStream_t stream = nullptr;
  crispy_mel* mel = nullptr;
  std::map<std::string, crispy::asr::Tensor> tensors;   // as named by the model file
  std::vector<float*> derived;             // fused / reordered copies owned by the handle
  size_t derived_bytes = 0;                // ... and their size (crispy_asr_memory_info)
  // resident quantised model (crispy_asr_load_resident): 2-D tensors stay as ggml blocks, de-quantised into ONE scratch
  // slot right in front of the kernel that consumes them (same stream: the consumer has finished before the next fill)
  bool resident = false;
  std::map<std::string, crispy::asr::QTensor> qtensors;
  void* q_scratch = nullptr;
  size_t q_scratch_bytes = 0;
  hipEvent_t ev_scratch = nullptr;           // orders a caller's stream against the handle's around the scratch slot
  const crispy::asr::QTensor* q_tok_emb = nullptr;
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
  std::vector<crispy::asr::EncLayer> enc;
  const float *tok_emb = nullptr, *dec_pos = nullptr, *dec_ln_w = nullptr, *dec_ln_b = nullptr;
  std::vector<crispy::asr::DecLayer> dec;
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
  crispy::TsState* d_ts_state = nullptr;             // [dcap_batch]
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
  crispy::BeamRow* d_beam_row = nullptr;             // [dcap_batch] a decoder's bookkeeping (beam_advance_kernel)
  int* d_beam_cand = nullptr;                // [3][dcap_batch][TS_MAX_CAND] the candidates of a step: ids, timestamp ids, log-probabilities
  int* d_beam_rec_parent = nullptr;          // [n_text_ctx][dcap_batch] the row a step's sequence came from
  double* d_beam_u = nullptr; size_t beam_u_bytes = 0;      // [max_new][rows][n_cand] the variates of a beam pass
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

namespace crispy {
namespace asr {

// ---- whisper_api.cpp ----
std::map<std::string, size_t> expected_tensors(const crispy_asr_hparams& hp);
// dense copy of a (row-concatenated) resident tensor in the handle's scratch slot, enqueued on `s` right in front of its consumer
int dq(crispy_asr* h, const QRef& r, bool f16, const float* gamma, hipStream_t s, const void** out);
void free_dec_ws(crispy_asr* h);
int reserve_enc(crispy_asr* h, int batch);
GemmArgs gemm(const float* A, long lda, const float* W, long ldw, float* C, long ldc, const float* bias, int M, int N, int K);

// ---- decode_steps.cpp ----
long logits_ld(const crispy_asr* h);
int reserve_dec(crispy_asr* h, int batch, int xclips = 0);
bool gemv_ref_ok(const QRef& r);
// rows of one group of fallback passes (whole clips x best_of); a grouping choice only -- every row's bits are those of
// its clip decoded alone.  The widest step the decode kernels take (SKINNY_MAX_M): a row costs less in a wide step (2.3 us at
// 512 rows against 2.6 at 128, Whisper-tiny), and 64 clips x best_of 5 are one group instead of three.
constexpr int kLadderRowsMax = 512;
struct Special {
  int sot, lang0, n_lang, n_lang_slots, translate, transcribe, solm, prev, nosp, not_, beg;
  bool multilingual;
};
Special vocab_specials(int n_vocab);
Special special_tokens(const crispy_asr* h);
int decode_ts(crispy_asr* h, const float* d_enc, int batch, const std::vector<std::vector<int>>& prompts, int rules,
              const int* seek, const int* seek_end, int max_new, const unsigned char* mask, const unsigned char* mask_first,
              float temperature, const double* u, int* tokens_out, int* tids_out, float* plog_out, float* nosp_out,
              int* n_out, int xgroup = 1);
double canonical(std::mt19937& g);
int decode_beam(crispy_asr* h, const float* d_enc, int n_clips, int n_dec, int n_cand, const std::vector<std::vector<int>>& clip_prompts,
                int rules, const int* seek, const int* seek_end, int max_new, const unsigned char* mask, const unsigned char* mask_first,
                float temperature, const std::vector<std::mt19937*>& rng, int* tokens_out, int* tids_out, float* plog_out,
                float* nosp_out, int* n_out);

// ---- whisper_full.cpp ----
int build_ts_masks(crispy_asr* h);

}  // namespace asr
}  // namespace crispy
