/*
 * crispy_hip.h -- C ABI of libcrispy_hip.so: the MI355X (gfx950) implementation of crispy's
 * audio compute hot path.  Plain pointers and sizes only; no C++/torch types cross this line.
 *
 * Every entry point names the reference interface it replaces (paths are into sleep3r/crispy):
 *
 *   RNNoise   nnnoiseless::DenoiseState::{new, process_frame}
 *               ctor       src-tauri/src/audio.rs:229
 *               call site  src-tauri/src/audio.rs:268   (480-sample f32 frames, int16 range)
 *               state reset on model hot-swap          src-tauri/src/audio.rs:942-967
 *   ASR       transcribe_rs::SpeechModel::transcribe (whisper_cpp::WhisperEngine)
 *               load       src-tauri/src/managers/transcription.rs:138-141
 *               call sites src-tauri/src/managers/transcription.rs:183-185, 213-215
 *
 * Conventions
 *   - every function returns CRISPY_OK (0) or a negative crispy_status; nothing throws or
 *     aborts across the boundary (the reference builds with panic=abort, Cargo.toml:10-20);
 *     crispy_last_error() returns a thread-local message for the last failure.
 *   - a handle is not re-entrant: the caller serialises calls on one handle, exactly as the
 *     reference does with Arc<Mutex<NsState>> (audio.rs:693) / Mutex<Option<engine>>
 *     (managers/transcription.rs:27).  Different handles may be used from different threads.
 *   - there is NO CPU fallback: without a gfx950 device every create/load call fails with
 *     CRISPY_ERR_NO_DEVICE.
 *   - ABI version: CRISPY_ABI_VERSION below is bumped whenever a struct grows, an entry point goes away or an argument
 *     changes meaning; a binding compares it with crispy_abi_version() when it loads the library and refuses a
 *     mismatch (a caller built against an older crispy_asr_opts would have the library read past its struct).
 *
 * Environment.  The release library reads exactly these three variables -- test hooks that choose between forms whose
 * results are bit-identical (the test that uses one asserts that); nothing else in a host's environment changes what runs,
 * and nothing in it changes a result.  Developer A/B knobs exist only in the `make dev` build, libcrispy_hip_dev.so
 * (crispy_amd/csrc/api_util.h: dev_env) -- among them CRISPY_ASR_DECODE=stages, the decode step as one launch per stage,
 * whose results are NOT bit-identical to the fused step kernels' (tests/test_gpu_fused_decode.py compares the two builds).
 *     CRISPY_RN_WAVES=1|3        frame-kernel form of a crispy_rn handle (default: 3 up to 1280 streams, 1 above);
 *                                read by crispy_rn_create*            (tests/test_gpu_rnnoise.py)
 *     CRISPY_ASR_PREFILL=seq     prompt one position per step instead of multi-position steps; read per decode call
 *                                                                     (tests/test_gpu_prefill.py)
 *     CRISPY_ASR_TILE_ROWS=192|256  tile height of the mode-1 encoder GEMMs; read once per process
 *                                                                     (tests/test_gpu_mode1.py)
 */
#ifndef CRISPY_HIP_H
#define CRISPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum crispy_status {
  CRISPY_OK = 0,
  CRISPY_ERR_INVALID_ARG = -1,
  CRISPY_ERR_NO_DEVICE = -2,
  CRISPY_ERR_HIP = -3,
  CRISPY_ERR_OOM = -4,
  CRISPY_ERR_BAD_MODEL = -5,
  CRISPY_ERR_UNSUPPORTED = -6,
  CRISPY_ERR_CANCELLED = -7 /* crispy_asr_transcribe_recording: the caller's cancel flag was set; no result */
} crispy_status;

#define CRISPY_RN_FRAME_SIZE 480      /* nnnoiseless::FRAME_SIZE (audio.rs:4) */
#define CRISPY_RN_WEIGHT_BYTES 87503  /* SURVEY.md Appendix A.5 */
#define CRISPY_RN_TAPS 72

/* Sample layouts of the batched frame tensors handed to crispy_rn_process*. */
typedef enum crispy_rn_layout {
  CRISPY_RN_LAYOUT_TBF = 0, /* [n_frames][n_streams][480]: one 10 ms tick of every stream is contiguous */
  CRISPY_RN_LAYOUT_BTF = 1  /* [n_streams][n_frames][480]: one stream's audio is contiguous */
} crispy_rn_layout;

typedef struct crispy_rn crispy_rn;

/* Message for the most recent failure on the calling thread ("" if none). */
const char *crispy_last_error(void);
/* "crispy_hip <version> gfx950" */
const char *crispy_version(void);
/* The ABI this header describes; crispy_abi_version() returns the one the library was built with.
 *   1: rounds 1 - 3.   2: round 4 (crispy_asr_opts 20 -> 44 bytes, crispy_asr_result gained windows, the staged RNNoise
 *   pipeline entry points removed).   3: round 5.   4: round 6 (this header: crispy_asr_transcribe_recording and its
 *   CRISPY_ERR_CANCELLED, the int16 sample transport crispy_rn_process_s16*). */
#define CRISPY_ABI_VERSION 4
int crispy_abi_version(void);
/* Number of usable gfx950 devices (0 when there is none; never fails). */
int crispy_device_count(void);
/* Self-test of the boundary's exception guard: every entry point is a function-try-block, so a C++ exception
 * (std::bad_alloc from a std::vector, std::system_error from a std::thread ...) becomes a status code and a
 * crispy_last_error() message instead of unwinding into a panic=abort host.  kind: 0 nothing, 1 std::bad_alloc,
 * 2 std::length_error, 3 std::runtime_error, 4 a non-standard exception, 5 a real failing std::vector::resize.
 * Returns what the guard made of it (CRISPY_ERR_OOM for 1, 2, 5; CRISPY_ERR_HIP for 3, 4; CRISPY_OK for 0). */
int crispy_selftest_exception_guard(int kind);

/*
 * DenoiseState::new() for n_streams independent streams (audio.rs:229).
 * weights: flat int8 blob of CRISPY_RN_WEIGHT_BYTES bytes, layer order input_dense, vad_gru,
 * vad_output, noise_gru, denoise_gru, denoise_output; per layer input weights [in][out]
 * (GRU [in][3N], gates z,r,h), recurrent weights [N][3N], bias.  The blob built into
 * nnnoiseless is not redistributable from here, so weights are always explicit.
 * All state starts at zero, as DenoiseState::new() does.
 */
int crispy_rn_create(const int8_t *weights, size_t nbytes, int n_streams, int device,
                     crispy_rn **out);
void crispy_rn_destroy(crispy_rn *h);

/*
 * Weights from an "rnnoise-nu model file version 1" text file -- the format nnnoiseless' RnnModel::from_read
 * parses [UPSTREAM-RECALL] -- so that a host without a weight blob of its own (the reference constructs
 * DenoiseState::new() with the model built into the crate, audio.rs:229; that blob cannot be redistributed from
 * here) can point the library at a model file instead.  crispy_rn_weights_from_file fills `blob`
 * (CRISPY_RN_WEIGHT_BYTES bytes, the layout crispy_rn_create takes) without touching a device; the topology is
 * fixed, so a file whose layer sizes or activation ids differ, that is truncated, or that holds a value outside
 * int8 is CRISPY_ERR_BAD_MODEL.  crispy_rn_create_from_file = parse + crispy_rn_create.
 */
int crispy_rn_weights_from_file(const char *path, int8_t *blob, size_t blob_bytes);
int crispy_rn_create_from_file(const char *path, int n_streams, int device, crispy_rn **out);

/* Fresh DenoiseState for one stream (stream >= 0) or all of them (stream == -1):
 * what audio.rs:955-965 does by replacing the processor. */
int crispy_rn_reset(crispy_rn *h, int stream);

int crispy_rn_n_streams(const crispy_rn *h);

/* Frames each rn_frame_kernel launch covers: a call of n frames is enqueued as ceil(n / this) launches
 * whose high-pass runs one launch ahead on a helper stream (bench.py's per-launch roofline uses it). */
int crispy_rn_frames_per_launch(void);
/* Number of rn_frame_kernel launches one call of n_frames makes: a segment (<= 250 frames) starts with short
 * sub-chunks (3, 4, 5, 7, 10 frames) so that the sequential high-pass of the first one is the only exposed one, then 12
 * (94 MB of high-passed signal at 4096 streams: still in the Infinity Cache when the frame kernel reads it). */
int crispy_rn_n_launches(int n_frames);

/*
 * process_frame for every stream, n_frames consecutive frames each (audio.rs:268).
 * in/out are HOST pointers to n_frames*n_streams*480 floats in `layout`; samples are f32 in
 * int16 range (the x32768 / /32768, clamp, volume and first-frame drop of audio.rs:261-278 stay
 * with the caller).  vad (nullable) receives the value process_frame returns,
 * [n_frames][n_streams].  Copies through a device staging buffer; returns when `out` is complete.
 * Calls above 8 MB are pipelined in pieces of frames: copy-in of piece i+1, the kernels of piece i and
 * copy-out of piece i-1 overlap (the copy-out side on its own host thread, because copies from pageable
 * memory block the calling thread), so a call costs about one direction of PCIe traffic.  Buffers registered
 * with crispy_host_register are copied by DMA without the runtime's staging.
 */
int crispy_rn_process(crispy_rn *h, const float *in, float *out, float *vad, int n_frames,
                      crispy_rn_layout layout);

/*
 * Page-lock (and later release) a host buffer the caller keeps across calls -- the ring buffers of
 * RnnNoiseProcessor (audio.rs:205-207) or a recording the transcriber reads -- so that crispy_rn_process and
 * crispy_asr_transcribe* copy it by DMA.  Thin wrappers over hipHostRegister / hipHostUnregister: a Rust host
 * does not need to link HIP for them.  Optional: unregistered (pageable) buffers work everywhere.
 */
int crispy_host_register(void *p, size_t bytes);
int crispy_host_unregister(void *p);

/*
 * Same, with DEVICE pointers (HBM-resident audio, no PCIe in the call).  Work is enqueued on
 * hip_stream (a hipStream_t, NULL = the handle's own stream) and the call returns without
 * waiting.  taps (nullable) receives CRISPY_RN_TAPS floats per frame and stream
 * [n_frames][n_streams][72]: features[42], gains[22], pitch_index, pitch_gain, vad, silence.
 */
int crispy_rn_process_device(crispy_rn *h, const float *d_in, float *d_out, float *d_vad,
                             float *d_taps, int n_frames, crispy_rn_layout layout,
                             void *hip_stream);

/*
 * Integer sample transport: the same call with int16 PCM in and out -- half the bytes of the f32 form across PCIe, and the
 * formats the reference's capture and recording paths actually hold (cpal i16 / u16 input streams, audio.rs:794-855; the
 * s16 WAV the transcriber reads back, commands/transcription.rs:306-313).
 *   in : sample s enters process_frame as (float)s -- the reference's s as f32 / 32768.0 (audio.rs:814) x 32768.0
 *        (audio.rs:264), both exact;  a u16 stream is the host's (s - 32768) first, as audio.rs:872 does.
 *   out: trunc(clamp(y / 32768, -1, 1) x 32767) for process_frame's output y -- the adapter's / 32768 and clamp
 *        (audio.rs:270-273, volume 1) followed by the WAV writer's quantisation, `(s.clamp(-1.0, 1.0) * 32767.0) as i16`
 *        (recording.rs:109-110), fused into the frame kernel's store.  Bit-exact with crispy_rn_process followed by that
 *        arithmetic on the host (tests/test_gpu_rnnoise.py).
 * State, layouts, vad, the first-frame drop (caller side) and the pipelining of large host calls are those of the f32
 * entry points; one handle may mix f32 and int16 calls.  Device pointers 16-byte aligned.
 */
int crispy_rn_process_s16(crispy_rn *h, const int16_t *in, int16_t *out, float *vad, int n_frames,
                          crispy_rn_layout layout);
int crispy_rn_process_s16_device(crispy_rn *h, const int16_t *d_in, int16_t *d_out, float *d_vad, int n_frames,
                                 crispy_rn_layout layout, void *hip_stream);

/* Block until everything enqueued on the handle's own stream has finished. */
int crispy_rn_synchronize(crispy_rn *h);

/*
 * Time the frame kernels of the next crispy_rn_process_device call with hipEvents recorded on
 * the launch stream (bench.py's roofline leg).  After that call and a synchronize,
 * crispy_rn_last_kernel_ms returns the device time of the dominant kernel (rn_frame_kernel) and
 * of the whole enqueue (high-pass + frame + history roll) in milliseconds.
 */
int crispy_rn_set_timing(crispy_rn *h, int enable);
int crispy_rn_last_kernel_ms(crispy_rn *h, float *frame_kernel_ms, float *total_ms);

/* Stage entry point (parity tests): the activation functions of the gain network exactly as the frame kernel
 * evaluates them (201-entry tanh table + interpolation, |x| >= 8 clamps; sigmoid != 0: 0.5 + 0.5 tansig(0.5 x)) on n
 * arbitrary f32 arguments, DEVICE pointers.  Bit-compared with the oracle over every table cell and both clamps. */
int crispy_rn_stage_tansig_device(crispy_rn *h, const float *d_x, float *d_y, size_t n, int sigmoid,
                                  void *hip_stream);

/* Developer aid for parity debugging: copy the per-stage debug capture of stream `stream`
 * for the LAST frame of the most recent call (layout mirrors oracle RNO_DBG_*). */
int crispy_rn_debug_capture(crispy_rn *h, int enable);
int crispy_rn_debug_read(crispy_rn *h, int stream, float *dst, size_t n_floats);

/* ------------------------------------------------------------------------------------------
 * Log-mel front end of the ASR path: whisper.cpp `log_mel_spectrogram`, the first stage of
 * transcribe_rs::SpeechModel::transcribe (managers/transcription.rs:183-185, 213-215).
 * 16 kHz f32 PCM in +-1 -> [n_mel][3000] f32 per clip (the frames the encoder consumes):
 * reflect-pad 200, periodic Hann 400, hop 160, power spectrum, mel filters, log10,
 * clamp to (clip max - 8), (x + 4) / 4.  
 * ------------------------------------------------------------------------------------------ */
#define CRISPY_MEL_FRAMES 3000
#define CRISPY_MEL_BINS 201

typedef struct crispy_mel crispy_mel;

/* filters: [n_mel][201] f32, the mel filter bank whisper.cpp reads from the model file
 * (n_mel = 80, or 128 for large-v3). */
int crispy_mel_create(const float *filters, int n_mel, int device, crispy_mel **out);
void crispy_mel_destroy(crispy_mel *h);

/* HOST pointers: pcm [batch][pcm_stride], n_samples[batch] (1..480000 each, the 30 s chunking of
 * commands/transcription.rs:249-302 stays with the caller), out [batch][n_mel][3000].
 * Returns when `out` is complete. */
int crispy_mel_compute(crispy_mel *h, const float *pcm, long pcm_stride, const int *n_samples,
                       int batch, float *out);

/* DEVICE pcm / outputs (n_samples stays a host array); enqueued on hip_stream (NULL = the handle's
 * stream) without waiting.  d_out [batch][n_mel][3000] and/or d_out_t [batch][3002][n_mel]
 * (frame-major, one zero frame of padding on both sides: the layout the encoder's first
 * convolution consumes); either may be NULL. */
int crispy_mel_compute_device(crispy_mel *h, const float *d_pcm, long pcm_stride,
                              const int *n_samples, int batch, float *d_out, float *d_out_t,
                              void *hip_stream);
/* Later 30 s windows of the clips of the LAST crispy_mel_compute_device call (whisper_full's seek loop): output
 * k is clip clip_idx[k] starting at mel frame seek[k] (0..3000; both host arrays), normalised with that clip's
 * maximum; frames past the end of the computed range are zero padding.  Same output layouts. */
int crispy_mel_window_device(crispy_mel *h, const int *clip_idx, const int *seek, int n, float *d_out,
                             float *d_out_t, void *hip_stream);
int crispy_mel_synchronize(crispy_mel *h);

/* ------------------------------------------------------------------------------------------
 * Whisper engine: replaces transcribe_rs::whisper_cpp::WhisperEngine behind `SpeechModel`
 *   load        managers/transcription.rs:138-141   WhisperEngine::load(&model_path)
 *   transcribe  managers/transcription.rs:183-185   engine.transcribe(&audio, &TranscribeOptions::default())
 * Architecture: SURVEY.md Appendix B.2 (pre-LN transformer, head dim 64, 1500 audio positions).
 * Round-1 numerics are f32 end to end on the f32-input matrix cores.
 * ------------------------------------------------------------------------------------------ */
typedef struct crispy_asr crispy_asr;

/* The hyper-parameter block of a whisper.cpp model file, minus ftype. */
typedef struct crispy_asr_hparams {
  int n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels;
} crispy_asr_hparams;

/* Model container.  Tensors are set by their model-file names ("encoder.conv1.weight",
 * "decoder.blocks.3.cross_attn.query.bias", ...) as f32 host arrays in PyTorch layout
 * ([out][in], conv [out][in][3]); crispy_asr_finalize checks completeness and builds the fused
 * device layouts.  mel_filters: [n_mels][201]. */
int crispy_asr_create(const crispy_asr_hparams *hp, const float *mel_filters, int device,
                      crispy_asr **out);
int crispy_asr_set_tensor(crispy_asr *h, const char *name, const float *data, size_t n_elems);
int crispy_asr_finalize(crispy_asr *h);
void crispy_asr_free(crispy_asr *h);
int crispy_asr_hparams_get(const crispy_asr *h, crispy_asr_hparams *out);

/* Stage entry points (parity tests): PCM (host, <= 30 s per clip) -> encoder output
 * [batch][1500][n_audio_state] (host), and the device-resident variant that starts from the
 * frame-major padded log-mel produced by crispy_mel_compute_device (d_out_t). */
int crispy_asr_encode(crispy_asr *h, const float *pcm, long pcm_stride, const int *n_samples,
                      int batch, float *out);
int crispy_asr_encode_device(crispy_asr *h, const float *d_mel_t, int batch, float *d_out,
                             void *hip_stream);
int crispy_asr_synchronize(crispy_asr *h);

/* Encoder GEMM operand precision.  0 (default): f32 operands on the f32-input matrix cores -- what the parity tests
 * against the float64 oracle pin to 1e-4.  1: f16 operands with f32 accumulation (weights stored as f16, activations
 * rounded to f16 on the way into LDS) -- the numerics of whisper.cpp's ggml matrix products [UPSTREAM-RECALL], on
 * v_mfma_f32_32x32x16_f16, activations that only feed a matrix product kept in f16, the whole convolution stem and the
 * encoder attention on the f16 matrix cores.  Decoder in this mode: cross K|V and the self-attention K|V cache kept in
 * f16 (as whisper.cpp's kv_self / kv_cross are), logits =
 * f16(LayerNorm(x)) . f16(token embedding)^T with f32 accumulation (whisper.cpp's f16 embedding under ggml's mul_mat),
 * and the projections with no LayerNorm in front of them (attention outputs, the MLP's second product) with f16
 * weights and the activation rounded to f16; the LayerNorm-folded projections (q | k | v, cross q, the MLP's first
 * product), LayerNorm and soft-max stay f32; GELU is ggml's (the f16-indexed table of the tanh form, evaluated on the fly).
 * oracle/whisper_oracle.py: encoder_forward_f16, DecoderCache(f16=True).
 * 2 (opt-in): mode 1 plus ggml's remaining rounding points [UPSTREAM-RECALL]: the decoder's LayerNorm output rounded to
 * f16 in front of q | k | v, cross q and the MLP's first product, which then run against f16 weights (the LayerNorm
 * becomes a launch of its own: three more launches per layer and step); and inside every attention -- encoder, decoder
 * self and cross -- the query rounded to f16 in front of K.q and the soft-max taken in full, normalised, and only then
 * rounded to f16 in front of P.V (the encoder runs a statistics pass over K first; a decode step two more hand-offs
 * between the waves of a row).  Not available for resident quantised models.
 * Oracle: encoder_forward_f16(attn16=True), DecoderCache(f16=True, ln16=True, attn16=True). */
int crispy_asr_set_precision(crispy_asr *h, int mode);

/* Stage entry point (parity tests): the last step of the decoder alone -- final LayerNorm and vocabulary projection
 * of d_x [batch][n_text_state] (device, f32) into d_logits [batch][n_vocab] (device, f32), in the current precision
 * mode.  Mode 1: LayerNorm output rounded to f16 against the f16 token embedding, f32 accumulation (ggml's mul_mat
 * over whisper.cpp's f16 `decoder.token_embedding.weight`).  batch <= 512. */
int crispy_asr_stage_logits_device(crispy_asr *h, const float *d_x, int batch, float *d_logits);

/* Greedy decoding (north_star: greedy; the sampling strategy transcribe-rs 0.3.11 selects is
 * unverifiable here).  Token ids listed with first_only = 0 are never
 * emitted; those with first_only = 1 only at the first sampled position (whisper's suppress_blank).
 * Nothing is suppressed until this is called. */
int crispy_asr_set_suppress(crispy_asr *h, const int *ids, int n, int first_only);

/* d_enc: DEVICE encoder output [batch][1500][n_text_state]; prompt: host token ids fed to every
 * clip (<|startoftranscript|>, language, <|transcribe|>, <|notimestamps|>); up to max_new tokens
 * are picked by argmax over f32 logits (ties: lowest id).  tokens_out [batch][max_new] (host),
 * n_out[b] = number of tokens before the first EOT (host, nullable), logits_out (host, nullable)
 * the logit of every pick. */
int crispy_asr_decode_greedy_device(crispy_asr *h, const float *d_enc, int batch, const int *prompt,
                                    int n_prompt, int max_new, int *tokens_out, int *n_out,
                                    float *logits_out);

/* SpeechModel::transcribe up to token ids: PCM (host) -> log-mel -> encoder -> greedy decoder.
 * batch == 0 is the empty-audio no-op of managers/transcription.rs:175-177.  Detokenisation needs
 * the vocabulary of a real model file and stays with the caller in this round. */
/* Same with a per-clip token for prompt position 1 (lang_tokens[batch], host, nullable). */
int crispy_asr_decode_greedy_lang_device(crispy_asr *h, const float *d_enc, int batch, const int *prompt,
                                         int n_prompt, const int *lang_tokens, int max_new,
                                         int *tokens_out, int *n_out, float *logits_out);

/* One decoding window per clip under the timestamp rules (whisper.cpp `whisper_process_logits`, which ports
 * openai-whisper's ApplyTimestampRules): <|notimestamps|> suppressed, timestamps in pairs except before EOT,
 * non-decreasing, first one <= 1.00 s, and a timestamp is forced once the probability mass of all timestamps
 * exceeds the most probable text token.  rules: 0 = whisper.cpp flavour [UPSTREAM-RECALL] (the window also
 * ends at a timestamp within 100 ms of seek_end), 1 = openai / HuggingFace flavour (forced first timestamp,
 * strictly later after a closed pair; pinned by tests/golden/whisper_tiny_ts_golden.npz).  prompt without
 * <|notimestamps|>; seek / seek_end [batch] (host, nullable): window start and audio length in mel frames.
 * tokens_out / tids_out [batch][max_new] (tids: the most probable timestamp token at every step, nullable),
 * n_out[b] = picks up to and including the one that ended the window.  Uses the crispy_asr_set_suppress masks. */
int crispy_asr_decode_timestamps_device(crispy_asr *h, const float *d_enc, int batch, const int *prompt,
                                        int n_prompt, const int *lang_tokens, int rules, const int *seek,
                                        const int *seek_end, int max_new, int *tokens_out, int *tids_out,
                                        int *n_out);

/* Stage entry point (parity tests): one pass of whisper_full's temperature ladder over one window per row -- what the
 * transcribe calls run per window and temperature.  Every row has its own prompt (prompts [rows][prompt_stride],
 * n_prompt[rows]; rows of different length decode in lock step, each bit-identical to the row decoded alone), the
 * suppression masks are whisper.cpp's own (specials never, " " and EOT not first).  u == NULL: greedy at temperature 0.
 * u [max_new][rows] (host doubles in [0, 1)): the sampling pick of whisper_sample_token at `temperature` > 0 -- logits /
 * temperature, std::discrete_distribution with u as its uniform variate.  Outputs [rows][max_new]: tokens, the most
 * probable timestamp token of every step (nullable), the log-probability of every pick (nullable; log-softmax over
 * everything allowed before the probability-mass rule); no_speech_prob_out [rows] (nullable); n_out as above. */
int crispy_asr_decode_window_device(crispy_asr *h, const float *d_enc, int rows, const int *prompts,
                                    const int *n_prompt, int prompt_stride, int rules, const int *seek,
                                    const int *seek_end, int max_new, float temperature, const double *u,
                                    int *tokens_out, int *tids_out, float *plog_out, float *no_speech_prob_out,
                                    int *n_out);

/* whisper.cpp language auto-detection: <|startoftranscript|> alone, arg-max over the language tokens.
 * lang_tokens_out[batch] (host).  English-only vocabularies -> CRISPY_ERR_UNSUPPORTED. */
int crispy_asr_detect_language_device(crispy_asr *h, const float *d_enc, int batch, int *lang_tokens_out);

int crispy_asr_transcribe_tokens(crispy_asr *h, const float *pcm, long pcm_stride,
                                 const int *n_samples, int batch, const int *prompt, int n_prompt,
                                 int max_new, int *tokens_out, int *n_out);

/* WhisperEngine::load(&model_path) (managers/transcription.rs:138-141): parse a whisper.cpp GGML
 * model file (hparams, mel filters, vocabulary, tensors) and build a finalized engine.  f32 / f16 tensors are
 * taken as they are; q4_0 / q4_1 / q5_0 / q5_1 / q8_0 blocks (catalog entries managers/model.rs:99,137) are
 * de-quantised to f32 at load time. */
int crispy_asr_load(const char *model_path, int device, crispy_asr **out);

/* The same, for the quantised files the reference's catalog ships (managers/model.rs:99 whisper-medium-q4_1.bin,
 * :137 ggml-large-v3-q5_0.bin; `size_mb` 492 / 1100 at :101, :139): the 2-D tensors STAY in HBM as the file's ggml
 * blocks (q4_0 / q4_1 / q5_0 / q5_1 / q8_0) and are de-quantised at the point of use -- to f16 operands in front of
 * every matrix product, f32 x gamma for the LayerNorm-folded decode projections -- into one scratch slot the size of the
 * largest layer matrix.  Resident weight bytes ~= file size (+ the token embedding once more as f16 in matrix-core
 * operand order for the logits).  The engine runs in precision mode 1 (whisper.cpp's f16-operand arithmetic); 
 * crispy_asr_set_precision(h, 0) is refused with CRISPY_ERR_UNSUPPORTED.  Results equal those of crispy_asr_load +
 * crispy_asr_set_precision(h, 1) on the same file bit for bit.  Only the matrices the engine consumes as blocks stay
 * quantised (attention and MLP weights, the token embedding); any other quantised tensor is inflated.  A file with no
 * quantised matrix at all (f32 / f16: ggml-small.bin, ggml-large-v3-turbo.bin, managers/model.rs:80,118) loads exactly
 * as crispy_asr_load does -- dense tensors, every precision mode available, crispy_asr_memory_info reports no quantised
 * bytes and no scratch slot.  crispy_asr_encode_device on a caller's stream is ordered against the handle's own stream
 * around the scratch slot (the encode starts after the work enqueued on the handle so far; the handle continues after it). */
int crispy_asr_load_resident(const char *model_path, int device, crispy_asr **out);

/* Device memory held by the model itself (not the per-call workspaces): every weight tensor, fused / folded / f16 copy
 * and resident quantised block (`weight_bytes`), the part of it that is quantised blocks (`quantised_bytes`), and the
 * de-quantisation scratch slot of a resident model (`scratch_bytes`).  Any out-pointer may be NULL. */
int crispy_asr_memory_info(const crispy_asr *h, size_t *weight_bytes, size_t *quantised_bytes, size_t *scratch_bytes);

/* Special-token ids of a whisper.cpp vocabulary of n_vocab entries [UPSTREAM-RECALL: whisper_vocab]: English-only
 * (51864): eot 50256, sot 50257, translate 50357, transcribe 50358, solm 50359, prev 50360, nosp 50361,
 * notimestamps 50362, first timestamp 50363, no language in the prompt; multilingual (51865, 51866 = large-v3):
 * everything one higher, and the tokens behind the language block one more per extra language.  Pure function,
 * no device needed. */
typedef struct crispy_asr_specials {
  int eot, sot, lang0, n_lang, translate, transcribe, solm, prev, nosp, notimestamps, beg, multilingual;
} crispy_asr_specials;
int crispy_asr_vocab_specials(int n_vocab, crispy_asr_specials *out);

/* Language token of an ISO code as whisper.cpp's `whisper_lang_id` table orders them [UPSTREAM-RECALL: g_lang, the
 * order of openai/whisper's LANGUAGES]: "en" -> lang0, "zh" -> lang0 + 1, ... "yue" -> lang0 + 99 (large-v3 only).
 * For hosts that hold the language as a string (`TranscribeOptions.language`); the result goes into
 * crispy_asr_opts::language_token.  English-only vocabularies have no language token: "en" gives 0 (= none),
 * anything else CRISPY_ERR_UNSUPPORTED; an unknown code, or one the vocabulary has no token for, CRISPY_ERR_INVALID_ARG.
 * "auto" and "" give 0 (auto-detect).  Pure function, no device needed. */
int crispy_asr_language_token(int n_vocab, const char *code, int *token_out);

/* Byte string of one vocabulary entry of a loaded model file (not NUL-terminated). */
int crispy_asr_token_text(const crispy_asr *h, int token, const char **text, size_t *len);

/* TranscribeOptions::default() (managers/transcription.rs:184): language unset, transcribe task, and whisper.cpp's
 * whisper_full_default_params(GREEDY) for everything else.  A zeroed struct IS that default: every field reads
 * "0 = whisper.cpp's default". */
typedef struct crispy_asr_opts {
  int language_token;  /* 0 = auto-detect (what TranscribeOptions::default() leaves to whisper.cpp); else the token id */
  int translate;       /* 0 = transcribe */
  int max_new_tokens;  /* 0 = whisper.cpp's per-window limit (n_text_ctx / 2 - 4; n_text_ctx / 2 without timestamps) */
  int no_timestamps;   /* 0 = whisper.cpp's default: timestamp tokens, 30 s windows advancing to the last closed
                          timestamp pair (whisper_full's seek loop), segments in the result.
                          1 = <|notimestamps|> prompt, one window, plain greedy arg-max, no segments, none of the
                          decision logic below. */
  int no_prev_text;    /* 0 = whisper.cpp's behaviour inside one whisper_full call [UPSTREAM-RECALL]: from the second
                          window on the decoder is conditioned on the text so far -- prompt <|startofprev|> + the last
                          <= n_text_ctx / 2 tokens of the previous windows (their timestamp tokens included) + the
                          usual <|startoftranscript|> ...; not when fewer than 5 s of audio are left, and not in a
                          re-decode at a temperature >= 0.5.  1 = every window starts from the bare prompt. */
  /* whisper_full's decision logic [UPSTREAM-RECALL: whisper_full_with_state, whisper_sequence_score].  A window is
   * decoded greedily at `temperature`; if its decoder failed (end of text before any timestamp away from the end of the
   * audio, no end within the token limit while less than half the window was covered, entropy of the last 32 tokens
   * below entropy_thold), or its average log-probability is below logprob_thold while no_speech_prob <
   * no_speech_thold, it is decoded again at temperature + temperature_inc, ... up to 1.0 -- above 0 with `best_of`
   * sampling decoders (std::mt19937(j) + std::discrete_distribution, decoder j seeded j at the start of every call),
   * the best-scoring one that did not fail wins; the last temperature is accepted as it is.  A window whose
   * no_speech_prob > no_speech_thold and whose average log-probability < logprob_thold yields no text. */
  float temperature;     /* first temperature of the ladder (0) */
  float temperature_inc; /* 0 = 0.2; < 0: no fallback (one pass at `temperature`, accepted as it is) */
  float entropy_thold;   /* 0 = 2.4; < 0: never fails on entropy */
  float logprob_thold;   /* 0 = -1.0 */
  float no_speech_thold; /* 0 = 0.6; >= 1: no window is ever dropped as silence */
  int best_of;           /* 0 = 5 */
  /* ---- appended with ABI 3 (zero / NULL = the behaviour before): the other whisper_full parameters a host may set ---- */
  int suppress_nst;      /* whisper_full_params.suppress_nst [UPSTREAM-RECALL]: 1 = the non-speech tokens -- whisper.cpp's list
                            of punctuation runs, brackets and music notes, each looked up in the MODEL'S vocabulary as it
                            stands and with a leading space, plus " -" and " '" -- are never picked.  Needs a model loaded
                            from a file (the vocabulary's text); a model without one has nothing to look up. */
  const int *initial_prompt;   /* whisper_full_params.prompt_tokens: token ids placed in FRONT of the conditioning text of every
                                  chunk of the call (whisper.cpp rotates them in front of prompt_past); the first window's prompt is
                                  then <|startofprev|> + the last <= n_text_ctx / 2 of them + the usual prompt.  The host
                                  tokenises (the reference's engine does: transcribe-rs hands whisper.cpp text).  NULL: none. */
  int n_initial_prompt;
  int carry_context;     /* 1 = whisper_full_params.no_context = false: the conditioning text the previous call on this handle
                            ended with is where this call's first window starts (a recording transcribed chunk by chunk
                            through one engine).  Single-chunk calls only (batch == 1); 0 = whisper.cpp's default
                            (no_context = true: every call starts clean). */
  int beam_size;         /* > 1: whisper.cpp's BEAM_SEARCH strategy [UPSTREAM-RECALL: whisper_full_with_state,
                            whisper_sample_token_topk] -- beam_size decoders per window at temperature 0 (best_of above), every
                            live decoder drawing beam_size candidate ids per step from its distribution, candidates sorted by
                            the sum of their log-probabilities and dealt to the decoders without repeating a sequence.  One
                            host round trip per token.  0 | 1: the GREEDY strategy.  At most 8 (WHISPER_MAX_DECODERS); needs
                            timestamps (no_timestamps = 1 with beam_size > 1: CRISPY_ERR_UNSUPPORTED). */
} crispy_asr_opts;

/* One segment of the result (managers/transcription.rs:223-233: `seg.start`, `seg.end`, `seg.text`),
 * seconds relative to the start of the chunk. */
typedef struct crispy_asr_segment {
  float t0, t1;
  const char *text;    /* UTF-8, NUL-terminated, untrimmed */
} crispy_asr_segment;

/* What whisper_full decided about one window of the seek loop (timestamp mode only). */
typedef struct crispy_asr_window {
  int seek;              /* window start, mel frames (10 ms) from the start of the chunk */
  int seek_advance;      /* frames the loop moved on by */
  int n_tokens;          /* tokens of this window that went into the result (0 when dropped) */
  int decoder;           /* which of the best_of decoders won (0 at temperature 0) */
  int failed;            /* 1 = the decoder failed and was accepted all the same (last temperature of the ladder) */
  int no_speech;         /* 1 = dropped by the no-speech rule */
  float temperature;     /* the temperature the accepted pass ran at */
  float no_speech_prob;  /* softmax of the last prompt position's unfiltered logits at <|nospeech|> (whisper.cpp takes it
                            there: the prompt pass yields the logits of its last position only) */
  float avg_logprob;     /* mean log-probability of the kept tokens (-inf for a decoder that failed before scoring) */
  float entropy;         /* entropy of the token histogram of the last 32 kept tokens */
} crispy_asr_window;

/* Library-owned result of one transcribe call; release with crispy_asr_free_result. */
typedef struct crispy_asr_result {
  const char *text;    /* UTF-8, NUL-terminated, untrimmed (the caller trims: transcription.rs:187) */
  const int *tokens;   /* the kept token ids (timestamp tokens included in timestamp mode), without <|endoftext|> */
  int n_tokens;
  int language_token;  /* the language token used (detected or given); 0 for English-only vocabularies */
  int n_segments;      /* 0 with no_timestamps (the reference then falls back to one segment, transcription.rs:236-249) */
  const crispy_asr_segment *segments;
  int n_windows;       /* windows of the seek loop, in order (0 with no_timestamps) */
  const crispy_asr_window *windows;
} crispy_asr_result;

/* engine.transcribe(&audio, &TranscribeOptions::default()) for ONE chunk of <= 480000 samples
 * (16 kHz f32, host).  n == 0 returns an empty result (transcription.rs:175-177), and so does a chunk of fewer than
 * 10 mel frames (100 ms; the frame count is 1 + (n - 200) / 160, i.e. 2999 for a full chunk): whisper.cpp's whisper_full
 * refuses it and yields no segments [UPSTREAM-RECALL: delta_min, n_len_org].  Needs a model
 * loaded from a file (vocabulary) for text; tokens are always returned.  opts == NULL is
 * TranscribeOptions::default(): language auto-detected, transcribe, timestamps on. */
int crispy_asr_transcribe(crispy_asr *h, const float *pcm16k, size_t n, const crispy_asr_opts *opts,
                          crispy_asr_result **out);
/* The same for `batch` chunks at once (pcm[i]: host pointer to n[i] <= 480000 samples, n[i] == 0 allowed):
 * one log-mel + encoder + language-detection + decoder pass over all clips (more than 512: in turns of 512).  results[batch]
 * receives one library-owned result per clip (free each); on failure every results[i] is NULL.  A clip's result is the result
 * of crispy_asr_transcribe on it alone, bit for bit, whatever the batch (every precision mode, every model width). */
int crispy_asr_transcribe_batch(crispy_asr *h, const float *const *pcm, const size_t *n, int batch,
                                const crispy_asr_opts *opts, crispy_asr_result **results);
void crispy_asr_free_result(crispy_asr_result *r);

/* `run_transcription`'s chunk loop (commands/transcription.rs:249-302, 363-400, 468) for a whole recording already at
 * 16 kHz: n samples are cut into 480 000-sample chunks (the last, partial one passed as it is), every chunk goes through
 * engine.transcribe(&chunk, opts), the chunk texts are trimmed (Rust's str::trim: Unicode white space), the non-blank ones
 * joined with ONE space -> result.text.  Where the reference's loop is serial (one chunk per engine call), the chunks here
 * are decoded in groups of `max_batch` (0 = 128) by one crispy_asr_transcribe_batch each -- they are independent:
 * TranscribeOptions::default() carries no context between chunks -- and the text is byte for byte what the chunk-by-chunk
 * loop gives (tests/test_gpu_recording.py).  Also in the result: the tokens of all chunks, the segments with their times
 * shifted by the chunk's start (chunk index x 30 s: `chunk_start_seconds`), the windows with `seek` shifted by 3000
 * frames per chunk; language_token = the first chunk's.
 *   cancel_flag (nullable): polled before every group and between the windows of the seek loop, as the reference polls its
 *     AtomicBool before every chunk (:251, :359, :402); once it reads non-zero the call returns CRISPY_ERR_CANCELLED and no
 *     result (the reference returns without saving anything).  The host sets it from another thread.
 *   progress (nullable): called on the calling thread after every group with (samples done, n, progress_user) -- what the
 *     reference turns into its "transcription-progress" event (:285-299).  It must not call into this handle.
 * n == 0: an empty result (:190-194).  opts->carry_context is refused (a single-chunk option). */
typedef void (*crispy_asr_progress_fn)(size_t samples_done, size_t samples_total, void *user);
int crispy_asr_transcribe_recording(crispy_asr *h, const float *pcm16k, size_t n, const crispy_asr_opts *opts,
                                    int max_batch, const volatile int *cancel_flag, crispy_asr_progress_fn progress,
                                    void *progress_user, crispy_asr_result **out);

/* ------------------------------------------------------------------------------------------
 * 48 kHz -> 16 kHz resampler between the denoiser and the ASR front end:
 * rubato FftFixedIn::<f32>::new(48000, 16000, 1024, 1, 1) as driven by
 * commands/transcription.rs:198-208, 314-357 (1024-sample chunks, last one zero-padded, the tail
 * shorter than one 1026-sample FFT block is never flushed), optionally preceded by the s16 WAV
 * hand-off of recording.rs:101-118 / commands/transcription.rs:306-313.
 * ------------------------------------------------------------------------------------------ */
typedef struct crispy_resampler crispy_resampler;
int crispy_resampler_create(int device, crispy_resampler **out);
void crispy_resampler_destroy(crispy_resampler *h);
/* 16 kHz samples produced for n_in 48 kHz samples: floor(ceil(n_in/1024)*1024 / 1026) * 342. */
long crispy_resampler_out_len(long n_in);
/* DEVICE pointers: d_in [batch][in_stride] (n_in valid samples each), d_out [batch][out_stride].
 * Every input sample is multiplied by `scale` first (1/32768 after the denoiser).  wav_s16: 0 = nothing
 * else, 1 = clamp(-1,1) (audio.rs:272), 2 = clamp, x32767 truncated to s16, /32768 (the WAV hand-off).
 * Enqueued on hip_stream (NULL = own stream). */
int crispy_resampler_process_device(crispy_resampler *h, const float *d_in, long in_stride, long n_in,
                                    int batch, float scale, int wav_s16, float *d_out,
                                    long out_stride, void *hip_stream);
int crispy_resampler_synchronize(crispy_resampler *h);

#ifdef __cplusplus
}
#endif
#endif /* CRISPY_HIP_H */
