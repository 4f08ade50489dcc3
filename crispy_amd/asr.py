"""Host-side mirror of the reference's ASR surface (transcribe-rs `SpeechModel::transcribe`,
reference call sites src-tauri/src/managers/transcription.rs:183-185), backed by the HIP library.

`LogMel` (whisper.cpp log_mel_spectrogram), `WhisperModel` (tensor-by-tensor container: encoder, greedy decoder,
language detection), `WhisperEngine` (GGML model file + text), and the reference's chunker / timestamp fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .mel_filters import whisper_mel_filters

N_FRAMES = 3000
CHUNK_SAMPLES = 480000  # 30 s at 16 kHz (commands/transcription.rs:175-176)


class LogMel:
    """whisper.cpp `log_mel_spectrogram` on the GPU: f32 PCM (16 kHz, +-1) -> [B, n_mel, 3000]."""

    def __init__(self, n_mel: int = 80, filters: np.ndarray | None = None, device: int = 0):
        f = whisper_mel_filters(n_mel) if filters is None else np.ascontiguousarray(filters, dtype=np.float32)
        if f.shape != (n_mel, 201):
            raise ValueError("filters must be [n_mel, 201]")
        self.n_mel = n_mel
        self._h = C.c_void_p()
        self._L = N.lib()        # the library this handle belongs to (a test may run another build side by side)
        self._ck(self._L.crispy_mel_create(f.ctypes.data, n_mel, device, C.byref(self._h)))

    def _ck(self, rc: int) -> None:
        N.check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.crispy_mel_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, clips) -> np.ndarray:
        """clips: a list of 1-D float32 arrays (each <= 480000 samples) or a 2-D array."""
        if isinstance(clips, np.ndarray) and clips.ndim == 2:
            lens = np.full(clips.shape[0], clips.shape[1], dtype=np.int32)
            pcm = np.ascontiguousarray(clips, dtype=np.float32)
        else:
            clips = [np.ascontiguousarray(c, dtype=np.float32).ravel() for c in clips]
            if not clips:
                return np.zeros((0, self.n_mel, N_FRAMES), np.float32)
            lens = np.array([c.size for c in clips], dtype=np.int32)
            pcm = np.zeros((len(clips), int(lens.max())), dtype=np.float32)
            for i, c in enumerate(clips):
                pcm[i, :c.size] = c
        out = np.empty((pcm.shape[0], self.n_mel, N_FRAMES), dtype=np.float32)
        self._ck(self._L.crispy_mel_compute(self._h, pcm.ctypes.data, pcm.shape[1], lens.ctypes.data,
                                           pcm.shape[0], out.ctypes.data))
        return out

    def compute_device(self, d_pcm: int, pcm_stride: int, n_samples: np.ndarray, d_out: int = 0, d_out_t: int = 0,
                       stream: int = 0):
        lens = np.ascontiguousarray(n_samples, dtype=np.int32)
        self._ck(self._L.crispy_mel_compute_device(self._h, d_pcm, pcm_stride, lens.ctypes.data, lens.size,
                                                  d_out or None, d_out_t or None, stream or None))

    def window_device(self, clip_idx, seek, d_out: int = 0, d_out_t: int = 0, stream: int = 0):
        """Later 30 s windows (whisper_full's seek loop) of the clips of the last compute_device call."""
        ci = np.ascontiguousarray(clip_idx, dtype=np.int32)
        sk = np.ascontiguousarray(seek, dtype=np.int32)
        self._ck(self._L.crispy_mel_window_device(self._h, ci.ctypes.data, sk.ctypes.data, ci.size, d_out or None,
                                                 d_out_t or None, stream or None))

    def synchronize(self):
        self._ck(self._L.crispy_mel_synchronize(self._h))


class WhisperModel:
    """Model container + engine on the GPU (`WhisperEngine::load` + `SpeechModel::transcribe`,
    managers/transcription.rs:138-141, 183-185).  Tensors are given by name in PyTorch layout."""

    def __init__(self, hp, weights: dict, filters: np.ndarray | None = None, device: int = 0):
        from .whisper_weights import tensor_shapes

        f = whisper_mel_filters(hp.n_mels) if filters is None else np.ascontiguousarray(filters, dtype=np.float32)
        self.hp = hp
        self._h = C.c_void_p()
        self._L = N.lib()        # the library this handle belongs to
        hpa = (C.c_int * 10)(*hp.as_ints())
        self._ck(self._L.crispy_asr_create(hpa, f.ctypes.data, device, C.byref(self._h)))
        for name, shape in tensor_shapes(hp).items():
            if name not in weights:
                raise KeyError(f"missing tensor {name}")
            w = np.ascontiguousarray(weights[name], dtype=np.float32)
            if w.shape != tuple(shape):
                raise ValueError(f"{name}: shape {w.shape}, expected {tuple(shape)}")
            self._ck(self._L.crispy_asr_set_tensor(self._h, name.encode(), w.ctypes.data, w.size))
        self._ck(self._L.crispy_asr_finalize(self._h))

    def _ck(self, rc: int) -> None:
        N.check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.crispy_asr_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _pack(clips):
        if isinstance(clips, np.ndarray) and clips.ndim == 2:
            return np.ascontiguousarray(clips, dtype=np.float32), np.full(clips.shape[0], clips.shape[1], np.int32)
        clips = [np.ascontiguousarray(c, dtype=np.float32).ravel() for c in clips]
        lens = np.array([c.size for c in clips], dtype=np.int32)
        pcm = np.zeros((len(clips), int(lens.max())), dtype=np.float32)
        for i, c in enumerate(clips):
            pcm[i, :c.size] = c
        return pcm, lens

    def encode(self, clips) -> np.ndarray:
        """PCM clips (16 kHz, <= 30 s each) -> encoder output [B, 1500, d]."""
        pcm, lens = self._pack(clips)
        out = np.empty((pcm.shape[0], self.hp.n_audio_ctx, self.hp.n_audio_state), dtype=np.float32)
        self._ck(self._L.crispy_asr_encode(self._h, pcm.ctypes.data, pcm.shape[1], lens.ctypes.data, pcm.shape[0],
                                          out.ctypes.data))
        return out

    def encode_device(self, d_mel_t: int, batch: int, d_out: int, stream: int = 0):
        self._ck(self._L.crispy_asr_encode_device(self._h, d_mel_t, batch, d_out, stream or None))

    def synchronize(self):
        self._ck(self._L.crispy_asr_synchronize(self._h))

    def memory_info(self) -> dict:
        """Device bytes held by the model itself: all weights / of which quantised blocks / de-quantisation scratch."""
        w, q, sc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._ck(self._L.crispy_asr_memory_info(self._h, C.byref(w), C.byref(q), C.byref(sc)))
        return {"weight_bytes": w.value, "quantised_bytes": q.value, "scratch_bytes": sc.value}

    def set_precision(self, mode: int):
        """0: f32 operands (default, the mode the oracle parity is pinned in); 1: f16 operands / f32 accumulation
        (whisper.cpp's ggml numerics); 2: mode 1 + the decoder's LayerNorm outputs rounded to f16 in front of q | k | v,
        cross q and fc1 (ggml's rounding points for those products too; opt-in)."""
        self._ck(self._L.crispy_asr_set_precision(self._h, int(mode)))

    def stage_logits_device(self, d_x: int, batch: int, d_logits: int):
        """Final LayerNorm + vocabulary projection of d_x [batch][n_text_state] into d_logits [batch][n_vocab] (device
        pointers), in the current precision mode: the last block of a decoder step, for parity tests."""
        self._ck(self._L.crispy_asr_stage_logits_device(self._h, d_x, batch, d_logits))

    def set_suppress(self, ids, first_only: bool = False):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._ck(self._L.crispy_asr_set_suppress(self._h, a.ctypes.data, a.size, int(first_only)))

    def set_default_suppression(self):
        """whisper.cpp's no-timestamps greedy masks: every id above <|endoftext|> is never emitted;
        blank (220) and EOT are not allowed as the first token [UPSTREAM-RECALL, SURVEY Appendix B.4]."""
        from .whisper_weights import TOK_EOT
        self.set_suppress(np.arange(TOK_EOT + 1, self.hp.n_vocab))
        self.set_suppress([220, TOK_EOT], first_only=True)

    def decode_greedy_device(self, d_enc: int, batch: int, prompt, max_new: int):
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        toks = np.empty((batch, max_new), dtype=np.int32)
        n = np.empty(batch, dtype=np.int32)
        lg = np.empty((batch, max_new), dtype=np.float32)
        self._ck(self._L.crispy_asr_decode_greedy_device(self._h, d_enc, batch, p.ctypes.data, p.size, max_new,
                                                        toks.ctypes.data, n.ctypes.data, lg.ctypes.data))
        return toks, n, lg

    def detect_language_device(self, d_enc: int, batch: int) -> np.ndarray:
        """Language token per clip (whisper.cpp auto-detection)."""
        out = np.empty(batch, dtype=np.int32)
        self._ck(self._L.crispy_asr_detect_language_device(self._h, d_enc, batch, out.ctypes.data))
        return out

    def decode_greedy_lang_device(self, d_enc: int, batch: int, prompt, lang_tokens, max_new: int):
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        lt = np.ascontiguousarray(lang_tokens, dtype=np.int32)
        toks = np.empty((batch, max_new), dtype=np.int32)
        n = np.empty(batch, dtype=np.int32)
        self._ck(self._L.crispy_asr_decode_greedy_lang_device(self._h, d_enc, batch, p.ctypes.data, p.size,
                                                             lt.ctypes.data, max_new, toks.ctypes.data, n.ctypes.data, None))
        return toks, n

    def decode_timestamps_device(self, d_enc: int, batch: int, prompt, max_new: int, rules: int = 0, lang_tokens=None,
                                 seek=None, seek_end=None):
        """One decoding window per clip under the timestamp rules (rules 0 = whisper.cpp, 1 = openai / HF flavour)
        -> (tokens [B, max_new], most probable timestamp token per step [B, max_new], picks per clip [B])."""
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        opt = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        lt, sk, se = opt(lang_tokens), opt(seek), opt(seek_end)
        ptr = lambda a: None if a is None else a.ctypes.data
        toks = np.empty((batch, max_new), dtype=np.int32)
        tids = np.empty((batch, max_new), dtype=np.int32)
        n = np.empty(batch, dtype=np.int32)
        self._ck(self._L.crispy_asr_decode_timestamps_device(self._h, d_enc, batch, p.ctypes.data, p.size, ptr(lt), int(rules),
                                                            ptr(sk), ptr(se), max_new, toks.ctypes.data, tids.ctypes.data,
                                                            n.ctypes.data))
        return toks, tids, n

    def decode_window_device(self, d_enc: int, prompts, max_new: int, rules: int = 0, seek=None, seek_end=None,
                             temperature: float = 0.0, u=None):
        """`crispy_asr_decode_window_device`: one pass of whisper_full's temperature ladder over one window per row; `prompts`
        is a list of token lists (one per row, lengths may differ), `u` [max_new, rows] float64 switches to the sampling
        pick at `temperature` -> (tokens, tids, plogs [rows, max_new], no_speech_prob [rows], picks per row)."""
        rows = len(prompts)
        stride = max(len(p) for p in prompts)
        pm = np.zeros((rows, stride), dtype=np.int32)
        for r, p in enumerate(prompts):
            pm[r, :len(p)] = p
        npr = np.asarray([len(p) for p in prompts], dtype=np.int32)
        opt = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        sk, se = opt(seek), opt(seek_end)
        uu = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        if uu is not None and uu.shape != (max_new, rows):
            raise ValueError(f"u must be [max_new, rows] = {(max_new, rows)}, got {uu.shape}")
        ptr = lambda a: None if a is None else a.ctypes.data
        toks = np.empty((rows, max_new), dtype=np.int32)
        tids = np.empty((rows, max_new), dtype=np.int32)
        plog = np.empty((rows, max_new), dtype=np.float32)
        nosp = np.empty(rows, dtype=np.float32)
        n = np.empty(rows, dtype=np.int32)
        self._ck(self._L.crispy_asr_decode_window_device(self._h, d_enc, rows, pm.ctypes.data, npr.ctypes.data, stride, int(rules),
                                                        ptr(sk), ptr(se), max_new, float(temperature), ptr(uu),
                                                        toks.ctypes.data, tids.ctypes.data, plog.ctypes.data, nosp.ctypes.data,
                                                        n.ctypes.data))
        return toks, tids, plog, nosp, n

    def transcribe_tokens(self, clips, prompt, max_new: int):
        """`SpeechModel::transcribe` up to token ids for a batch of <= 30 s clips."""
        if len(clips) == 0:
            return np.zeros((0, max_new), np.int32), np.zeros(0, np.int32)
        pcm, lens = self._pack(clips)
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        toks = np.empty((pcm.shape[0], max_new), dtype=np.int32)
        n = np.empty(pcm.shape[0], dtype=np.int32)
        self._ck(self._L.crispy_asr_transcribe_tokens(self._h, pcm.ctypes.data, pcm.shape[1], lens.ctypes.data,
                                                     pcm.shape[0], p.ctypes.data, p.size, max_new,
                                                     toks.ctypes.data, n.ctypes.data))
        return toks, n


class WhisperEngine(WhisperModel):
    """`WhisperEngine::load(&model_path)` + `transcribe(&audio, &TranscribeOptions::default())`
    (managers/transcription.rs:138-141, 183-185) over a whisper.cpp GGML model file."""

    def __init__(self, model_path: str, device: int = 0, resident: bool = False):  # noqa: super().__init__ is the tensor-by-tensor path
        """resident=True: `crispy_asr_load_resident` -- a quantised file (the catalog's q4_1 / q5_0 models,
        managers/model.rs:99,137) keeps its matrices as ggml blocks in HBM and runs in precision mode 1 only."""
        from .whisper_weights import HParams

        self._h = C.c_void_p()
        self._L = N.lib()
        load = self._L.crispy_asr_load_resident if resident else self._L.crispy_asr_load
        self._ck(load(str(model_path).encode(), device, C.byref(self._h)))
        hpa = (C.c_int * 10)()
        self._ck(self._L.crispy_asr_hparams_get(self._h, hpa))
        self.hp = HParams(*[int(v) for v in hpa])

    def token_text(self, token: int) -> bytes:
        p, n = C.c_char_p(), C.c_size_t()
        self._ck(self._L.crispy_asr_token_text(self._h, int(token), C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)

    def transcribe(self, audio: np.ndarray, max_new_tokens: int = 0, translate: bool = False, language_token: int = 0,
                   timestamps: bool = False, prev_text: bool = True, **decision):
        """One chunk (<= 480000 samples at 16 kHz) -> (text, token ids); empty audio -> ("", []).
        language_token = 0 auto-detects, as `TranscribeOptions::default()` does.  timestamps = True is whisper.cpp's
        default decoding mode (timestamp tokens, seek loop); its segments are kept in `self.last_segments` as
        (t0 seconds, t1 seconds, text), what whisper_full decided per window in `self.last_windows` (dicts of the
        `crispy_asr_window` fields).  prev_text = True: from the second window of the seek loop on, the decoder is
        conditioned on the text so far (whisper.cpp's prompt_past).  decision: temperature, temperature_inc,
        entropy_thold, logprob_thold, no_speech_thold, best_of (0 / absent = whisper.cpp's default; fallback=False is
        shorthand for temperature_inc = -1: one greedy pass per window)."""
        a = np.ascontiguousarray(audio, dtype=np.float32).ravel()
        opts = make_opts(language_token, translate, max_new_tokens, timestamps, prev_text, **decision)
        res = C.c_void_p()
        self._ck(self._L.crispy_asr_transcribe(self._h, a.ctypes.data if a.size else None, a.size, C.byref(opts),
                                              C.byref(res)))
        try:
            text, tokens, self.last_language_token, self.last_segments, self.last_windows = _read_result(res)
        finally:
            self._L.crispy_asr_free_result(res)
        return text, tokens

    def transcribe_segments(self, audio: np.ndarray, max_new_tokens: int = 0, language_token: int = 0, prev_text: bool = True,
                            **decision):
        """`engine.transcribe(..)` with whisper.cpp's default options -> (text, [(t0, t1, text)], token ids)."""
        text, tokens = self.transcribe(audio, max_new_tokens, False, language_token, timestamps=True, prev_text=prev_text,
                                       **decision)
        return text, self.last_segments, tokens


def make_opts(language_token=0, translate=False, max_new_tokens=0, timestamps=False, prev_text=True, fallback=True,
              temperature=0.0, temperature_inc=0.0, entropy_thold=0.0, logprob_thold=0.0, no_speech_thold=0.0, best_of=0,
              suppress_nst=False, initial_prompt=None, carry_context=False, beam_size=0):
    """`crispy_asr_opts`; the decision fields read "0 = whisper.cpp's default" (include/crispy_hip.h).  The struct keeps a
    reference to the initial-prompt array (`_keep`) for as long as it lives."""
    if not fallback and temperature_inc == 0.0:
        temperature_inc = -1.0
    o = N.AsrOpts(int(language_token), int(translate), int(max_new_tokens), 0 if timestamps else 1, 0 if prev_text else 1,
                  float(temperature), float(temperature_inc), float(entropy_thold), float(logprob_thold),
                  float(no_speech_thold), int(best_of))
    o.suppress_nst = 1 if suppress_nst else 0
    if initial_prompt is not None and len(initial_prompt):
        arr = (C.c_int * len(initial_prompt))(*[int(t) for t in initial_prompt])
        o._keep = arr
        o.initial_prompt = C.cast(arr, C.POINTER(C.c_int))
        o.n_initial_prompt = len(initial_prompt)
    o.carry_context = 1 if carry_context else 0
    o.beam_size = int(beam_size)
    return o


def _read_result(res) -> tuple:
    r = C.cast(res, C.POINTER(N.AsrResult)).contents
    text = r.text.decode("utf-8", "replace") if r.text else ""
    tokens = [int(r.tokens[i]) for i in range(r.n_tokens)]
    segs = [(float(r.segments[i].t0), float(r.segments[i].t1), (r.segments[i].text or b"").decode("utf-8", "replace"))
            for i in range(r.n_segments)]
    wins = [{k: getattr(r.windows[i], k) for k, _ in N.AsrWindow._fields_} for i in range(r.n_windows)]
    return text, tokens, int(r.language_token), segs, wins


def transcribe_batch(engine: "WhisperEngine", clips, max_new_tokens: int = 0, language_token: int = 0,
                     timestamps: bool = False, with_segments: bool = False, prev_text: bool = True, **decision):
    """`crispy_asr_transcribe_batch`: a list of chunks (each <= 30 s, empty allowed) -> [(text, tokens, language)]
    (+ segments and window decisions with with_segments)."""
    arrs = [np.ascontiguousarray(c, dtype=np.float32).ravel() for c in clips]
    nb = len(arrs)
    ptrs = (C.c_void_p * max(nb, 1))(*[a.ctypes.data if a.size else None for a in arrs])
    lens = (C.c_size_t * max(nb, 1))(*[a.size for a in arrs])
    res = (C.c_void_p * max(nb, 1))()
    opts = make_opts(language_token, False, max_new_tokens, timestamps, prev_text, **decision)
    engine._ck(engine._L.crispy_asr_transcribe_batch(engine._h, ptrs, lens, nb, C.byref(opts), res))
    out = []
    for i in range(nb):
        r = C.c_void_p(res[i])
        try:
            full = _read_result(r)
            out.append(full if with_segments else full[:3])
        finally:
            engine._L.crispy_asr_free_result(r)
    return out


def transcribe_recording_serial(engine: "WhisperEngine", pcm16k: np.ndarray, max_new_tokens: int = 0, timestamps: bool = False,
                                **decision) -> str:
    """The chunker of `run_transcription` as the reference runs it (commands/transcription.rs:249-302, 363-400, 468): hard
    30 s cuts, ONE engine call per chunk, the final partial chunk passed as is, chunk texts trimmed and joined with a single
    space.  Kept as the statement `transcribe_recording` is tested against."""
    parts = []
    x = np.ascontiguousarray(pcm16k, dtype=np.float32).ravel()
    for t0 in range(0, x.size, CHUNK_SAMPLES):
        text, _ = engine.transcribe(x[t0:t0 + CHUNK_SAMPLES], max_new_tokens, timestamps=timestamps, **decision)
        text = text.strip()
        if text:
            parts.append(text)
    return " ".join(parts)


class Cancelled(Exception):
    """`transcribe_recording` saw its cancel flag set (the reference returns without saving: transcription.rs:251,359,402)."""


def transcribe_recording(engine: "WhisperEngine", pcm16k: np.ndarray, max_new_tokens: int = 0, timestamps: bool = False,
                         max_batch: int = 0, cancel=None, progress=None, with_result: bool = False, **decision):
    """`crispy_asr_transcribe_recording`: the same chunk loop with the chunks decoded side by side, `max_batch` (0 = 128) per
    engine call -- an hour of audio is one call instead of 120 -- and the reference's two hooks:
    cancel: a `ctypes.c_int` another thread (or the progress callback) sets non-zero -> raises `Cancelled`;
    progress: callable (samples_done, samples_total), called after every group of chunks (transcription.rs:285-299).
    Returns the text; with_result also (text, tokens, language, segments, windows) of the whole recording."""
    x = np.ascontiguousarray(pcm16k, dtype=np.float32).ravel()
    opts = make_opts(0, False, max_new_tokens, timestamps, True, **decision)
    res = C.c_void_p()
    cb = N.PROGRESS_FN(lambda done, total, _user: progress(int(done), int(total))) if progress else N.PROGRESS_FN()
    rc = engine._L.crispy_asr_transcribe_recording(engine._h, x.ctypes.data if x.size else None, x.size, C.byref(opts), int(max_batch),
                                                   C.byref(cancel) if cancel is not None else None, cb, None, C.byref(res))
    if rc == N.ERR_CANCELLED:
        raise Cancelled()
    engine._ck(rc)
    try:
        full = _read_result(res)
    finally:
        engine._L.crispy_asr_free_result(res)
    return full if with_result else full[0]


def transcribe_with_timestamps(engine: "WhisperEngine", audio: np.ndarray, chunk_offset_seconds: float,
                               max_new_tokens: int = 0, use_segments: bool = False):
    """`TranscriptionManager::transcribe_with_timestamps` (managers/transcription.rs:200-249): empty audio or empty
    trimmed text -> []; with segments from the engine (use_segments, :223-240) every non-blank segment shifted by
    the chunk offset, text untrimmed; otherwise (:242-248, what the reference reports for Whisper) one segment
    spanning the chunk, (offset, offset + len/16000, trimmed text)."""
    a = np.ascontiguousarray(audio, dtype=np.float32).ravel()
    if a.size == 0:
        return []
    text, _ = engine.transcribe(a, max_new_tokens, timestamps=use_segments)
    text = text.strip()
    if not text:
        return []
    if use_segments and engine.last_segments:
        return [(float(chunk_offset_seconds) + t0, float(chunk_offset_seconds) + t1, st)
                for t0, t1, st in engine.last_segments if st.strip()]
    return [(float(chunk_offset_seconds), float(chunk_offset_seconds) + a.size / 16000.0, text)]
