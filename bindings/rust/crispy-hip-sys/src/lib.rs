//! Rust side of `libcrispy_hip.so` (C ABI: `include/crispy_hip.h`) for sleep3r/crispy.
//!
//! * the `extern "C"` block declares every entry point of the header, one to one (checked mechanically against
//!   the header by `tests/test_rust_binding_matches_header.py`, because this crate cannot be compiled where the
//!   library is built: no cargo / rustc in that image);
//! * [`DenoiseState`] has the surface of `nnnoiseless::DenoiseState` that crispy uses
//!   (`src-tauri/src/audio.rs:229` `DenoiseState::new()`, `audio.rs:268` `process_frame(&mut out, &in) -> f32`), so
//!   `RnnNoiseProcessor` (`audio.rs:202-315`) swaps one `use` line;
//! * [`GpuWhisperEngine`] has the surface of `transcribe_rs::whisper_cpp::WhisperEngine` behind `SpeechModel`
//!   (`src-tauri/src/managers/transcription.rs:138-141` load, `:183-185` / `:213-215` transcribe).
//!
//! Nothing here panics across the boundary and nothing unwinds out of it: every entry point of the library is a
//! function-try-block that turns C++ exceptions into status codes (the reference builds with `panic = "abort"`,
//! `src-tauri/Cargo.toml:10-20`).
#![allow(non_camel_case_types)]

use std::ffi::{c_char, c_float, c_int, c_long, c_uchar, c_void, CStr, CString};
use std::path::Path;

// ------------------------------------------------------------------------------------------------------------
// raw declarations (include/crispy_hip.h)
// ------------------------------------------------------------------------------------------------------------
pub const CRISPY_OK: c_int = 0;
pub const CRISPY_ERR_INVALID_ARG: c_int = -1;
pub const CRISPY_ERR_NO_DEVICE: c_int = -2;
pub const CRISPY_ERR_HIP: c_int = -3;
pub const CRISPY_ERR_OOM: c_int = -4;
pub const CRISPY_ERR_BAD_MODEL: c_int = -5;
pub const CRISPY_ERR_UNSUPPORTED: c_int = -6;
pub const CRISPY_ERR_CANCELLED: c_int = -7;

pub const CRISPY_RN_FRAME_SIZE: usize = 480;
pub const CRISPY_RN_WEIGHT_BYTES: usize = 87503;
pub const CRISPY_RN_TAPS: usize = 72;
/// The ABI this file was written against (crispy_hip.h: CRISPY_ABI_VERSION); every constructor checks it.
pub const CRISPY_ABI_VERSION: c_int = 4;
pub const CRISPY_MEL_FRAMES: usize = 3000;
pub const CRISPY_MEL_BINS: usize = 201;

/// `crispy_rn_layout`: C enums are `int`-sized on every ABI this library targets.
pub type crispy_rn_layout = c_int;
pub const CRISPY_RN_LAYOUT_TBF: crispy_rn_layout = 0;
pub const CRISPY_RN_LAYOUT_BTF: crispy_rn_layout = 1;

#[repr(C)]
pub struct crispy_rn {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crispy_mel {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crispy_asr {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crispy_resampler {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct crispy_asr_hparams {
    pub n_vocab: c_int,
    pub n_audio_ctx: c_int,
    pub n_audio_state: c_int,
    pub n_audio_head: c_int,
    pub n_audio_layer: c_int,
    pub n_text_ctx: c_int,
    pub n_text_state: c_int,
    pub n_text_head: c_int,
    pub n_text_layer: c_int,
    pub n_mels: c_int,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct crispy_asr_specials {
    pub eot: c_int,
    pub sot: c_int,
    pub lang0: c_int,
    pub n_lang: c_int,
    pub translate: c_int,
    pub transcribe: c_int,
    pub solm: c_int,
    pub prev: c_int,
    pub nosp: c_int,
    pub notimestamps: c_int,
    pub beg: c_int,
    pub multilingual: c_int,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct crispy_asr_opts {
    pub language_token: c_int,
    pub translate: c_int,
    pub max_new_tokens: c_int,
    pub no_timestamps: c_int,
    pub no_prev_text: c_int,
    /// whisper_full's decision logic; every field reads "0 = whisper.cpp's default" (include/crispy_hip.h), so
    /// `crispy_asr_opts::default()` is `TranscribeOptions::default()`.
    pub temperature: c_float,
    pub temperature_inc: c_float,
    pub entropy_thold: c_float,
    pub logprob_thold: c_float,
    pub no_speech_thold: c_float,
    pub best_of: c_int,
    /// ABI 3: whisper_full_params.suppress_nst, prompt_tokens / prompt_n_tokens, !no_context, and beam_search.beam_size
    /// (> 1: the BEAM_SEARCH strategy, at most 8 decoders).
    pub suppress_nst: c_int,
    pub initial_prompt: *const c_int,
    pub n_initial_prompt: c_int,
    pub carry_context: c_int,
    pub beam_size: c_int,
}
impl Default for crispy_asr_opts {
    /// All zero / NULL = `TranscribeOptions::default()` (a raw pointer has no derived Default).
    fn default() -> Self {
        // SAFETY: every field is an integer, a float or a raw pointer: the all-zero bit pattern is a valid value of each.
        unsafe { std::mem::zeroed() }
    }
}

#[repr(C)]
pub struct crispy_asr_segment {
    pub t0: c_float,
    pub t1: c_float,
    pub text: *const c_char,
}

/// What whisper_full decided about one window of the seek loop.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct crispy_asr_window {
    pub seek: c_int,
    pub seek_advance: c_int,
    pub n_tokens: c_int,
    pub decoder: c_int,
    pub failed: c_int,
    pub no_speech: c_int,
    pub temperature: c_float,
    pub no_speech_prob: c_float,
    pub avg_logprob: c_float,
    pub entropy: c_float,
}

#[repr(C)]
pub struct crispy_asr_result {
    pub text: *const c_char,
    pub tokens: *const c_int,
    pub n_tokens: c_int,
    pub language_token: c_int,
    pub n_segments: c_int,
    pub segments: *const crispy_asr_segment,
    pub n_windows: c_int,
    pub windows: *const crispy_asr_window,
}

/// `crispy_asr_progress_fn`: called after every group of chunks with (samples done, samples total, user).
pub type crispy_asr_progress_fn = Option<unsafe extern "C" fn(samples_done: usize, samples_total: usize, user: *mut c_void)>;

extern "C" {
    pub fn crispy_last_error() -> *const c_char;
    pub fn crispy_version() -> *const c_char;
    pub fn crispy_abi_version() -> c_int;
    pub fn crispy_device_count() -> c_int;
    pub fn crispy_selftest_exception_guard(kind: c_int) -> c_int;

    pub fn crispy_rn_create(weights: *const i8, nbytes: usize, n_streams: c_int, device: c_int, out: *mut *mut crispy_rn) -> c_int;
    pub fn crispy_rn_destroy(h: *mut crispy_rn);
    pub fn crispy_rn_weights_from_file(path: *const c_char, blob: *mut i8, blob_bytes: usize) -> c_int;
    pub fn crispy_rn_create_from_file(path: *const c_char, n_streams: c_int, device: c_int, out: *mut *mut crispy_rn) -> c_int;
    pub fn crispy_rn_reset(h: *mut crispy_rn, stream: c_int) -> c_int;
    pub fn crispy_rn_n_streams(h: *const crispy_rn) -> c_int;
    pub fn crispy_rn_frames_per_launch() -> c_int;
    pub fn crispy_rn_n_launches(n_frames: c_int) -> c_int;
    pub fn crispy_rn_process(h: *mut crispy_rn, input: *const c_float, output: *mut c_float, vad: *mut c_float, n_frames: c_int, layout: crispy_rn_layout) -> c_int;
    pub fn crispy_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn crispy_host_unregister(p: *mut c_void) -> c_int;
    pub fn crispy_rn_process_device(h: *mut crispy_rn, d_in: *const c_float, d_out: *mut c_float, d_vad: *mut c_float, d_taps: *mut c_float, n_frames: c_int, layout: crispy_rn_layout, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_rn_process_s16(h: *mut crispy_rn, input: *const i16, out: *mut i16, vad: *mut c_float, n_frames: c_int, layout: crispy_rn_layout) -> c_int;
    pub fn crispy_rn_process_s16_device(h: *mut crispy_rn, d_in: *const i16, d_out: *mut i16, d_vad: *mut c_float, n_frames: c_int, layout: crispy_rn_layout, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_rn_synchronize(h: *mut crispy_rn) -> c_int;
    pub fn crispy_rn_set_timing(h: *mut crispy_rn, enable: c_int) -> c_int;
    pub fn crispy_rn_last_kernel_ms(h: *mut crispy_rn, frame_kernel_ms: *mut c_float, total_ms: *mut c_float) -> c_int;
    pub fn crispy_rn_stage_tansig_device(h: *mut crispy_rn, d_x: *const c_float, d_y: *mut c_float, n: usize, sigmoid: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_rn_debug_capture(h: *mut crispy_rn, enable: c_int) -> c_int;
    pub fn crispy_rn_debug_read(h: *mut crispy_rn, stream: c_int, dst: *mut c_float, n_floats: usize) -> c_int;

    pub fn crispy_mel_create(filters: *const c_float, n_mel: c_int, device: c_int, out: *mut *mut crispy_mel) -> c_int;
    pub fn crispy_mel_destroy(h: *mut crispy_mel);
    pub fn crispy_mel_compute(h: *mut crispy_mel, pcm: *const c_float, pcm_stride: c_long, n_samples: *const c_int, batch: c_int, out: *mut c_float) -> c_int;
    pub fn crispy_mel_compute_device(h: *mut crispy_mel, d_pcm: *const c_float, pcm_stride: c_long, n_samples: *const c_int, batch: c_int, d_out: *mut c_float, d_out_t: *mut c_float, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_mel_window_device(h: *mut crispy_mel, clip_idx: *const c_int, seek: *const c_int, n: c_int, d_out: *mut c_float, d_out_t: *mut c_float, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_mel_synchronize(h: *mut crispy_mel) -> c_int;

    pub fn crispy_asr_create(hp: *const crispy_asr_hparams, mel_filters: *const c_float, device: c_int, out: *mut *mut crispy_asr) -> c_int;
    pub fn crispy_asr_set_tensor(h: *mut crispy_asr, name: *const c_char, data: *const c_float, n_elems: usize) -> c_int;
    pub fn crispy_asr_finalize(h: *mut crispy_asr) -> c_int;
    pub fn crispy_asr_free(h: *mut crispy_asr);
    pub fn crispy_asr_hparams_get(h: *const crispy_asr, out: *mut crispy_asr_hparams) -> c_int;
    pub fn crispy_asr_encode(h: *mut crispy_asr, pcm: *const c_float, pcm_stride: c_long, n_samples: *const c_int, batch: c_int, out: *mut c_float) -> c_int;
    pub fn crispy_asr_encode_device(h: *mut crispy_asr, d_mel_t: *const c_float, batch: c_int, d_out: *mut c_float, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_asr_synchronize(h: *mut crispy_asr) -> c_int;
    pub fn crispy_asr_set_precision(h: *mut crispy_asr, mode: c_int) -> c_int;
    pub fn crispy_asr_stage_logits_device(h: *mut crispy_asr, d_x: *const f32, batch: c_int, d_logits: *mut f32) -> c_int;
    pub fn crispy_asr_set_suppress(h: *mut crispy_asr, ids: *const c_int, n: c_int, first_only: c_int) -> c_int;
    pub fn crispy_asr_decode_greedy_device(h: *mut crispy_asr, d_enc: *const c_float, batch: c_int, prompt: *const c_int, n_prompt: c_int, max_new: c_int, tokens_out: *mut c_int, n_out: *mut c_int, logits_out: *mut c_float) -> c_int;
    pub fn crispy_asr_decode_greedy_lang_device(h: *mut crispy_asr, d_enc: *const c_float, batch: c_int, prompt: *const c_int, n_prompt: c_int, lang_tokens: *const c_int, max_new: c_int, tokens_out: *mut c_int, n_out: *mut c_int, logits_out: *mut c_float) -> c_int;
    pub fn crispy_asr_decode_timestamps_device(h: *mut crispy_asr, d_enc: *const c_float, batch: c_int, prompt: *const c_int, n_prompt: c_int, lang_tokens: *const c_int, rules: c_int, seek: *const c_int, seek_end: *const c_int, max_new: c_int, tokens_out: *mut c_int, tids_out: *mut c_int, n_out: *mut c_int) -> c_int;
    pub fn crispy_asr_detect_language_device(h: *mut crispy_asr, d_enc: *const c_float, batch: c_int, lang_tokens_out: *mut c_int) -> c_int;
    pub fn crispy_asr_transcribe_tokens(h: *mut crispy_asr, pcm: *const c_float, pcm_stride: c_long, n_samples: *const c_int, batch: c_int, prompt: *const c_int, n_prompt: c_int, max_new: c_int, tokens_out: *mut c_int, n_out: *mut c_int) -> c_int;
    pub fn crispy_asr_load(model_path: *const c_char, device: c_int, out: *mut *mut crispy_asr) -> c_int;
    pub fn crispy_asr_load_resident(model_path: *const c_char, device: c_int, out: *mut *mut crispy_asr) -> c_int;
    pub fn crispy_asr_memory_info(h: *const crispy_asr, weight_bytes: *mut usize, quantised_bytes: *mut usize, scratch_bytes: *mut usize) -> c_int;
    pub fn crispy_asr_vocab_specials(n_vocab: c_int, out: *mut crispy_asr_specials) -> c_int;
    pub fn crispy_asr_language_token(n_vocab: c_int, code: *const c_char, token_out: *mut c_int) -> c_int;
    pub fn crispy_asr_token_text(h: *const crispy_asr, token: c_int, text: *mut *const c_char, len: *mut usize) -> c_int;
    pub fn crispy_asr_decode_window_device(h: *mut crispy_asr, d_enc: *const c_float, rows: c_int, prompts: *const c_int, n_prompt: *const c_int, prompt_stride: c_int, rules: c_int, seek: *const c_int, seek_end: *const c_int, max_new: c_int, temperature: c_float, u: *const f64, tokens_out: *mut c_int, tids_out: *mut c_int, plog_out: *mut c_float, no_speech_prob_out: *mut c_float, n_out: *mut c_int) -> c_int;
    pub fn crispy_asr_transcribe(h: *mut crispy_asr, pcm16k: *const c_float, n: usize, opts: *const crispy_asr_opts, out: *mut *mut crispy_asr_result) -> c_int;
    pub fn crispy_asr_transcribe_batch(h: *mut crispy_asr, pcm: *const *const c_float, n: *const usize, batch: c_int, opts: *const crispy_asr_opts, results: *mut *mut crispy_asr_result) -> c_int;
    pub fn crispy_asr_free_result(r: *mut crispy_asr_result);
    pub fn crispy_asr_transcribe_recording(h: *mut crispy_asr, pcm16k: *const c_float, n: usize, opts: *const crispy_asr_opts, max_batch: c_int, cancel_flag: *const c_int, progress: crispy_asr_progress_fn, progress_user: *mut c_void, out: *mut *mut crispy_asr_result) -> c_int;

    pub fn crispy_resampler_create(device: c_int, out: *mut *mut crispy_resampler) -> c_int;
    pub fn crispy_resampler_destroy(h: *mut crispy_resampler);
    pub fn crispy_resampler_out_len(n_in: c_long) -> c_long;
    pub fn crispy_resampler_process_device(h: *mut crispy_resampler, d_in: *const c_float, in_stride: c_long, n_in: c_long, batch: c_int, scale: c_float, wav_s16: c_int, d_out: *mut c_float, out_stride: c_long, hip_stream: *mut c_void) -> c_int;
    pub fn crispy_resampler_synchronize(h: *mut crispy_resampler) -> c_int;
}

// ------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------
/// Status code + the library's thread-local message for it.
#[derive(Debug, Clone)]
pub struct CrispyError {
    pub code: c_int,
    pub message: String,
}
impl std::fmt::Display for CrispyError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "crispy_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for CrispyError {}

fn check(rc: c_int) -> Result<(), CrispyError> {
    if rc == CRISPY_OK {
        return Ok(());
    }
    // SAFETY: crispy_last_error returns a NUL-terminated thread-local buffer that lives as long as the thread.
    let message = unsafe { CStr::from_ptr(crispy_last_error()) }.to_string_lossy().into_owned();
    Err(CrispyError { code: rc, message })
}

/// Refuses a library built from another header: a `crispy_asr_opts` of another size would be read past its end.
fn check_abi() -> Result<(), CrispyError> {
    // SAFETY: no arguments, no state.
    let got = unsafe { crispy_abi_version() };
    if got != CRISPY_ABI_VERSION {
        return Err(CrispyError { code: CRISPY_ERR_UNSUPPORTED, message: format!("libcrispy_hip ABI {got}, binding written for {CRISPY_ABI_VERSION}") });
    }
    Ok(())
}

fn path_cstring(p: &Path) -> Result<CString, CrispyError> {
    CString::new(p.to_string_lossy().as_bytes()).map_err(|_| CrispyError {
        code: CRISPY_ERR_INVALID_ARG,
        message: "path contains a NUL byte".into(),
    })
}

// ------------------------------------------------------------------------------------------------------------
// nnnoiseless-shaped denoiser
// ------------------------------------------------------------------------------------------------------------
/// == `nnnoiseless::FRAME_SIZE` (audio.rs:4)
pub const FRAME_SIZE: usize = CRISPY_RN_FRAME_SIZE;

/// Same shape as `nnnoiseless::DenoiseState`: a constructor and `process_frame(out, in) -> f32`, one stream.
///
/// `RnnNoiseProcessor` keeps everything it does around the call: x32768 (audio.rs:264), /32768 + clamp + volume
/// (audio.rs:270-273), first-frame drop (audio.rs:275-278).
pub struct DenoiseState {
    h: *mut crispy_rn,
    failed: Option<CrispyError>,
}
// SAFETY: the handle is owned exclusively and the library allows different threads to use a handle as long as calls
// are serialised -- which `&mut self` guarantees; the reference moves the processor into the cpal closure
// (audio.rs:751) behind Arc<Mutex<..>> (audio.rs:693).
unsafe impl Send for DenoiseState {}

impl DenoiseState {
    /// `DenoiseState::new()` has no arguments upstream because the model is compiled into the crate.  That blob
    /// cannot be redistributed with this library, so the model comes from an rnnoise-nu text file (the format
    /// `nnnoiseless::RnnModel::from_read` parses) ...
    pub fn from_model_file(path: &Path) -> Result<Box<Self>, CrispyError> {
        check_abi()?;
        let c = path_cstring(path)?;
        let mut h = std::ptr::null_mut();
        // SAFETY: c outlives the call; h is a valid out-pointer.
        check(unsafe { crispy_rn_create_from_file(c.as_ptr(), 1, 0, &mut h) })?;
        Ok(Box::new(Self { h, failed: None }))
    }
    /// ... or from the flat 87 503-byte int8 blob (layer order input_dense, vad_gru, vad_output, noise_gru,
    /// denoise_gru, denoise_output).
    pub fn from_weights(weights: &[i8]) -> Result<Box<Self>, CrispyError> {
        check_abi()?;
        let mut h = std::ptr::null_mut();
        // SAFETY: the slice is valid for weights.len() bytes; the library copies it before returning.
        check(unsafe { crispy_rn_create(weights.as_ptr(), weights.len(), 1, 0, &mut h) })?;
        Ok(Box::new(Self { h, failed: None }))
    }
    /// `process_frame(&mut self, output, input) -> f32` (audio.rs:268): 480 samples each, f32 in int16 range;
    /// returns the VAD probability.  The signature is infallible like upstream's (slice lengths are asserted, as upstream
    /// does).  A device failure after construction -- which upstream cannot have -- must not turn into permanent
    /// silence in the audio callback with nobody told: the frame is passed through UNDENOISED (the microphone keeps
    /// working), VAD 0 is returned, and the error is latched for the host to poll with `take_error()` (e.g. once per
    /// second from the thread that owns `NsState`, which can then fall back to `NoiseSuppressionMode::Off`).
    pub fn process_frame(&mut self, output: &mut [f32], input: &[f32]) -> f32 {
        assert_eq!(input.len(), FRAME_SIZE);
        assert_eq!(output.len(), FRAME_SIZE);
        let mut vad = 0f32;
        // SAFETY: both slices hold exactly one frame; the call returns when `output` is complete.
        let rc = unsafe { crispy_rn_process(self.h, input.as_ptr(), output.as_mut_ptr(), &mut vad, 1, CRISPY_RN_LAYOUT_TBF) };
        if rc != CRISPY_OK {
            if self.failed.is_none() {
                self.failed = check(rc).err();       // the first failure, with the library's message for this thread
            }
            output.copy_from_slice(input);
            return 0.0;
        }
        vad
    }
    /// The first error `process_frame` met since the last call, if any (and clears it).
    pub fn take_error(&mut self) -> Option<CrispyError> {
        self.failed.take()
    }
    /// True once a `process_frame` call has failed and the error has not been taken yet.
    pub fn has_failed(&self) -> bool {
        self.failed.is_some()
    }
    /// What `set_monitoring_model` does by replacing the processor (audio.rs:955-965).
    pub fn reset(&mut self) -> Result<(), CrispyError> {
        // SAFETY: valid handle.
        check(unsafe { crispy_rn_reset(self.h, -1) })
    }
}
impl Drop for DenoiseState {
    fn drop(&mut self) {
        // SAFETY: the handle came from crispy_rn_create* and is destroyed exactly once.
        unsafe { crispy_rn_destroy(self.h) }
    }
}

/// B streams in lock step: `[n_frames][n_streams][480]` tensors from host memory (a server denoising thousands of
/// calls, or a long recording cut into independent segments).  This is where the GPU pays off.
pub struct BatchDenoiser {
    h: *mut crispy_rn,
    n_streams: usize,
}
unsafe impl Send for BatchDenoiser {}
impl BatchDenoiser {
    pub fn from_model_file(path: &Path, n_streams: usize, device: i32) -> Result<Self, CrispyError> {
        let c = path_cstring(path)?;
        let mut h = std::ptr::null_mut();
        check(unsafe { crispy_rn_create_from_file(c.as_ptr(), n_streams as c_int, device, &mut h) })?;
        Ok(Self { h, n_streams })
    }
    /// input / output: `n_frames * n_streams * 480` samples, frame-major (`CRISPY_RN_LAYOUT_TBF`); vad (optional):
    /// `n_frames * n_streams`.
    pub fn process(&mut self, input: &[f32], output: &mut [f32], vad: Option<&mut [f32]>, n_frames: usize) -> Result<(), CrispyError> {
        let n = n_frames * self.n_streams * FRAME_SIZE;
        if input.len() != n || output.len() != n || vad.as_ref().map_or(false, |v| v.len() != n_frames * self.n_streams) {
            return Err(CrispyError { code: CRISPY_ERR_INVALID_ARG, message: "BatchDenoiser::process: slice lengths".into() });
        }
        let vp = vad.map_or(std::ptr::null_mut(), |v| v.as_mut_ptr());
        check(unsafe { crispy_rn_process(self.h, input.as_ptr(), output.as_mut_ptr(), vp, n_frames as c_int, CRISPY_RN_LAYOUT_TBF) })
    }
}
impl Drop for BatchDenoiser {
    fn drop(&mut self) {
        unsafe { crispy_rn_destroy(self.h) }
    }
}

// ------------------------------------------------------------------------------------------------------------
// transcribe-rs shaped engine
// ------------------------------------------------------------------------------------------------------------
/// One segment of a transcript: seconds relative to the start of the chunk (managers/transcription.rs:223-233).
#[derive(Debug, Clone, PartialEq)]
pub struct Segment {
    pub start: f32,
    pub end: f32,
    pub text: String,
}
/// What `engine.transcribe(..)` returns as far as the reference reads it (`.text`, `.segments`).
#[derive(Debug, Clone, Default, PartialEq)]
pub struct Transcript {
    pub text: String,
    pub segments: Option<Vec<Segment>>,
    pub tokens: Vec<i32>,
    pub language_token: i32,
    /// Per window of whisper_full's seek loop: temperature accepted, no_speech_prob, avg_logprob, entropy, dropped / failed.
    pub windows: Vec<crispy_asr_window>,
}

/// `transcribe_rs::whisper_cpp::WhisperEngine`: `load(&path)` + `transcribe(&audio, &TranscribeOptions::default())`.
pub struct GpuWhisperEngine {
    h: *mut crispy_asr,
}
// SAFETY: as for DenoiseState; the reference keeps the engine in Mutex<Option<Box<dyn SpeechModel>>>
// (managers/transcription.rs:27) and calls it from a per-request thread (commands/transcription.rs:63).
unsafe impl Send for GpuWhisperEngine {}

impl GpuWhisperEngine {
    /// `WhisperEngine::load(&model_path)` (managers/transcription.rs:138-141): a whisper.cpp GGML model file.
    pub fn load(model_path: &Path) -> Result<Self, CrispyError> {
        check_abi()?;
        let c = path_cstring(model_path)?;
        let mut h = std::ptr::null_mut();
        // resident load: a quantised catalog file (managers/model.rs:99,137) stays quantised in HBM (file-sized) and runs in
        // precision mode 1 only; an f32 / f16 file (ggml-small.bin, large-v3-turbo) loads exactly as crispy_asr_load loads it
        // -- dense, every precision mode available.  Either way the engine is in precision mode 1 afterwards
        check(unsafe { crispy_asr_load_resident(c.as_ptr(), 0, &mut h) })?;
        // whisper.cpp, the engine this one stands in for, multiplies f16 operands with f32 accumulation and keeps its
        // K|V caches in f16: precision mode 1 is that arithmetic (and 2.3 x the f32 mode's speed).  The library's own
        // default stays f32 -- the mode its 1e-4 parity against the float64 oracle is stated in.
        let engine = Self { h };
        check(unsafe { crispy_asr_set_precision(engine.h, 1) })?;
        Ok(engine)
    }
    /// Vocabulary size of the loaded model (51864 English-only, 51865 multilingual, 51866 large-v3).
    pub fn n_vocab(&self) -> c_int {
        let mut hp = crispy_asr_hparams::default();
        if unsafe { crispy_asr_hparams_get(self.h, &mut hp) } == CRISPY_OK { hp.n_vocab } else { 0 }
    }
    /// 0: exact f32 products (parity mode; refused with CRISPY_ERR_UNSUPPORTED for a quantised file, which `load` keeps
    /// resident as ggml blocks); 1: whisper.cpp's f16-operand arithmetic (the default of `load`); 2: mode 1 plus ggml's
    /// remaining rounding points (LayerNorm outputs of the decode step, queries and normalised probabilities inside every
    /// attention), opt-in and slower.
    pub fn set_precision(&mut self, mode: i32) -> Result<(), CrispyError> {
        check(unsafe { crispy_asr_set_precision(self.h, mode as c_int) })
    }
    /// One chunk of at most 480 000 samples (16 kHz, f32 in +-1); `opts = None` is `TranscribeOptions::default()`:
    /// language auto-detected, transcribe task, timestamps on.  Empty audio gives an empty transcript
    /// (managers/transcription.rs:175-177).
    pub fn transcribe_chunk(&mut self, audio: &[f32], opts: Option<&crispy_asr_opts>) -> Result<Transcript, CrispyError> {
        let mut r: *mut crispy_asr_result = std::ptr::null_mut();
        let o = opts.map_or(std::ptr::null(), |o| o as *const crispy_asr_opts);
        let p = if audio.is_empty() { std::ptr::null() } else { audio.as_ptr() };
        check(unsafe { crispy_asr_transcribe(self.h, p, audio.len(), o, &mut r) })?;
        Ok(unsafe { take_result(r) })
    }
    /// `run_transcription`'s chunk loop (commands/transcription.rs:249-302, 363-400, 468) over a whole 16 kHz recording in
    /// ONE call: 30 s chunks decoded `max_batch` at a time (0 = 128), chunk texts trimmed and joined with a space.
    /// `cancel`: the reference's `Arc<AtomicBool>` (polled before every group and between windows; a set flag gives
    /// `Err` with `code == CRISPY_ERR_CANCELLED`, where the reference returns without saving).  `progress(done, total)` in
    /// samples after every group -- what the reference emits as "transcription-progress" (:285-299).
    pub fn transcribe_recording(&mut self, audio: &[f32], opts: Option<&crispy_asr_opts>, max_batch: usize,
                                cancel: Option<&std::sync::atomic::AtomicBool>, mut progress: Option<&mut dyn FnMut(usize, usize)>)
                                -> Result<Transcript, CrispyError> {
        unsafe extern "C" fn tramp(done: usize, total: usize, user: *mut c_void) {
            // SAFETY: `user` is the `&mut Option<&mut dyn FnMut>` of the enclosing call, alive for its whole duration.
            let f = &mut *(user as *mut Option<&mut dyn FnMut(usize, usize)>);
            if let Some(f) = f.as_mut() { f(done, total) }
        }
        // AtomicBool has the layout of a u8; the library reads an int.  A relay thread-free way: mirror the flag into an
        // AtomicI32 the callback refreshes would miss cancels between groups, so the flag the library polls is an i32 the
        // host sets directly when it can (`cancel_i32`); for an AtomicBool the mirror is refreshed by a watcher thread.
        let mirror = std::sync::Arc::new(std::sync::atomic::AtomicI32::new(0));
        let stop = std::sync::Arc::new(std::sync::atomic::AtomicBool::new(false));
        let watcher = cancel.map(|c| {
            let (m, st) = (mirror.clone(), stop.clone());
            let c_ptr = c as *const std::sync::atomic::AtomicBool as usize;
            std::thread::spawn(move || {
                // SAFETY: the AtomicBool outlives this call (borrowed for it); the thread is joined before it returns.
                let c = unsafe { &*(c_ptr as *const std::sync::atomic::AtomicBool) };
                while !st.load(std::sync::atomic::Ordering::Relaxed) {
                    if c.load(std::sync::atomic::Ordering::Relaxed) { m.store(1, std::sync::atomic::Ordering::Relaxed); break; }
                    std::thread::sleep(std::time::Duration::from_millis(2));
                }
            })
        });
        let mut r: *mut crispy_asr_result = std::ptr::null_mut();
        let o = opts.map_or(std::ptr::null(), |o| o as *const crispy_asr_opts);
        let p = if audio.is_empty() { std::ptr::null() } else { audio.as_ptr() };
        let cb: crispy_asr_progress_fn = if progress.is_some() { Some(tramp) } else { None };
        let rc = unsafe {
            crispy_asr_transcribe_recording(self.h, p, audio.len(), o, max_batch as c_int, mirror.as_ptr() as *const c_int, cb,
                                            &mut progress as *mut Option<&mut dyn FnMut(usize, usize)> as *mut c_void, &mut r)
        };
        stop.store(true, std::sync::atomic::Ordering::Relaxed);
        if let Some(w) = watcher { let _ = w.join(); }
        check(rc)?;
        Ok(unsafe { take_result(r) })
    }
}

/// Copies a library-owned result out and frees it.
/// SAFETY: `r` is a result a successful transcribe call returned and nobody has freed.
unsafe fn take_result(r: *mut crispy_asr_result) -> Transcript {
    let res = &*r;
    let text = if res.text.is_null() { String::new() } else { CStr::from_ptr(res.text).to_string_lossy().into_owned() };
    let tokens = if res.n_tokens > 0 { std::slice::from_raw_parts(res.tokens, res.n_tokens as usize).to_vec() } else { Vec::new() };
    let segments = if res.n_segments > 0 {
        Some(std::slice::from_raw_parts(res.segments, res.n_segments as usize).iter().map(|s| Segment {
            start: s.t0,
            end: s.t1,
            text: if s.text.is_null() { String::new() } else { CStr::from_ptr(s.text).to_string_lossy().into_owned() },
        }).collect())
    } else {
        None
    };
    let windows = if res.n_windows > 0 { std::slice::from_raw_parts(res.windows, res.n_windows as usize).to_vec() } else { Vec::new() };
    let out = Transcript { text, segments, tokens, language_token: res.language_token, windows };
    crispy_asr_free_result(r);
    out
}
impl Drop for GpuWhisperEngine {
    fn drop(&mut self) {
        unsafe { crispy_asr_free(self.h) }
    }
}

/// `impl SpeechModel for GpuWhisperEngine`, so that `TranscriptionManager::load_model`
/// (managers/transcription.rs:137-141) can box it as `LoadedEngine` for `EngineType::Whisper`.
/// The trait's exact item list lives in transcribe-rs 0.3.11, which is not vendored with the reference; the shape
/// below is what the reference's call sites require of it (`transcribe(&mut self, &[f32], &TranscribeOptions) ->
/// Result<R, E: Display>` with `R.text: String`, `R.segments: Option<Vec<S>>`, `S.{start, end, text}`:
/// managers/transcription.rs:183-187, 223-233).
#[cfg(feature = "speech-model")]
mod speech_model {
    use super::*;
    use transcribe_rs::{SpeechModel, TranscribeOptions, TranscriptionResult, TranscriptionSegment};

    impl SpeechModel for GpuWhisperEngine {
        // `TranscribeOptions { language: Option<String>, translate: bool }` [UPSTREAM-RECALL: transcribe-rs 0.3.11; the
        // reference only ever passes `TranscribeOptions::default()`, managers/transcription.rs:184,214 -- language unset,
        // transcribe]: a set language becomes its token, an unset one is detected per chunk, as whisper.cpp does.
        // (What transcribe-rs may set on whisper.cpp beyond these two -- suppress_nst, an initial prompt, no_context = false --
        // has a field in `crispy_asr_opts` since ABI 3; a host that knows the engine's settings passes them through
        // `transcribe_chunk(audio, Some(&opts))`; so does `beam_size` for an engine configured for BeamSearch.)
        fn transcribe(&mut self, audio: &[f32], options: &TranscribeOptions) -> Result<TranscriptionResult, Box<dyn std::error::Error + Send + Sync>> {
            let mut o = crispy_asr_opts::default();
            if let Some(code) = options.language.as_deref() {
                let c = std::ffi::CString::new(code)?;
                let mut tok: c_int = 0;
                check(unsafe { crispy_asr_language_token(self.n_vocab(), c.as_ptr(), &mut tok) })?;
                o.language_token = tok;
            }
            o.translate = options.translate as c_int;
            let t = self.transcribe_chunk(audio, Some(&o))?;
            Ok(TranscriptionResult {
                text: t.text,
                segments: t.segments.map(|v| v.into_iter().map(|s| TranscriptionSegment { start: s.start, end: s.end, text: s.text }).collect()),
            })
        }
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn exception_guard_is_status_codes() {
        // needs no device
        unsafe {
            assert_eq!(crispy_selftest_exception_guard(0), CRISPY_OK);
            assert_eq!(crispy_selftest_exception_guard(1), CRISPY_ERR_OOM);
            assert_eq!(crispy_selftest_exception_guard(3), CRISPY_ERR_HIP);
        }
    }

    #[test]
    fn specials_of_the_three_vocabularies() {
        let mut sp = crispy_asr_specials::default();
        unsafe { assert_eq!(crispy_asr_vocab_specials(51864, &mut sp), CRISPY_OK) };
        assert_eq!((sp.eot, sp.sot, sp.translate, sp.notimestamps, sp.beg), (50256, 50257, 50357, 50362, 50363));
        unsafe { assert_eq!(crispy_asr_vocab_specials(51865, &mut sp), CRISPY_OK) };
        assert_eq!((sp.eot, sp.sot, sp.translate, sp.notimestamps, sp.beg), (50257, 50258, 50358, 50363, 50364));
    }

    #[test]
    fn missing_model_file_is_an_error_not_a_panic() {
        let e = DenoiseState::from_model_file(Path::new("/nonexistent/model.txt")).err().unwrap();
        assert_eq!(e.code, CRISPY_ERR_BAD_MODEL);
    }
}
