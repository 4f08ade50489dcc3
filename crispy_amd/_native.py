"""ctypes binding of crispy_amd/libcrispy_hip.so (the C ABI declared in include/crispy_hip.h).

There is no Python or CPU fallback: if the shared library is missing, or no gfx950 device is
present, the failure is raised to the caller."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRISPY_HIP_LIB: developer override to A/B another build of the same HIP library (tools/ab_variants.sh)
LIB_PATH = os.environ.get("CRISPY_HIP_LIB") or os.path.join(_HERE, "libcrispy_hip.so")

RN_FRAME = 480
ABI_VERSION = 4
RN_WEIGHT_BYTES = 87503
RN_TAPS = 72
RN_DBG_FLOATS = 4304
LAYOUT_TBF = 0
LAYOUT_BTF = 1

# every symbol include/crispy_hip.h declares (checked by tests/test_abi.py)
RN_SYMBOLS = (
    "crispy_last_error", "crispy_version", "crispy_abi_version", "crispy_device_count",
    "crispy_rn_create", "crispy_rn_destroy", "crispy_rn_reset", "crispy_rn_n_streams",
    "crispy_rn_frames_per_launch", "crispy_rn_n_launches",
    "crispy_rn_process", "crispy_rn_process_device", "crispy_rn_process_s16", "crispy_rn_process_s16_device", "crispy_rn_synchronize",
    "crispy_rn_set_timing", "crispy_rn_last_kernel_ms",
    "crispy_rn_debug_capture", "crispy_rn_debug_read", "crispy_rn_stage_tansig_device",
    "crispy_host_register", "crispy_host_unregister",
    "crispy_rn_weights_from_file", "crispy_rn_create_from_file", "crispy_selftest_exception_guard",
)


MEL_SYMBOLS = ("crispy_mel_create", "crispy_mel_destroy", "crispy_mel_compute",
               "crispy_mel_compute_device", "crispy_mel_window_device", "crispy_mel_synchronize")
ASR_SYMBOLS = ("crispy_asr_create", "crispy_asr_set_tensor", "crispy_asr_finalize", "crispy_asr_free",
               "crispy_asr_hparams_get", "crispy_asr_encode", "crispy_asr_encode_device", "crispy_asr_synchronize",
               "crispy_asr_set_suppress", "crispy_asr_decode_greedy_device", "crispy_asr_transcribe_tokens",
               "crispy_asr_load", "crispy_asr_load_resident", "crispy_asr_memory_info", "crispy_asr_token_text", "crispy_asr_transcribe", "crispy_asr_free_result",
               "crispy_asr_decode_greedy_lang_device", "crispy_asr_detect_language_device",
               "crispy_asr_transcribe_batch", "crispy_asr_decode_timestamps_device", "crispy_asr_set_precision",
               "crispy_asr_vocab_specials", "crispy_asr_stage_logits_device", "crispy_asr_language_token",
               "crispy_asr_decode_window_device", "crispy_asr_transcribe_recording")
RS_SYMBOLS = ("crispy_resampler_create", "crispy_resampler_destroy", "crispy_resampler_out_len",
              "crispy_resampler_process_device", "crispy_resampler_synchronize")
ALL_SYMBOLS = RN_SYMBOLS + MEL_SYMBOLS + ASR_SYMBOLS + RS_SYMBOLS

class AsrSpecials(C.Structure):
    """crispy_asr_specials"""
    _fields_ = [(k, C.c_int) for k in ("eot", "sot", "lang0", "n_lang", "translate", "transcribe", "solm", "prev",
                                       "nosp", "notimestamps", "beg", "multilingual")]


class AsrOpts(C.Structure):
    """crispy_asr_opts"""
    _fields_ = [("language_token", C.c_int), ("translate", C.c_int), ("max_new_tokens", C.c_int),
                ("no_timestamps", C.c_int), ("no_prev_text", C.c_int),
                ("temperature", C.c_float), ("temperature_inc", C.c_float), ("entropy_thold", C.c_float),
                ("logprob_thold", C.c_float), ("no_speech_thold", C.c_float), ("best_of", C.c_int),
                ("suppress_nst", C.c_int), ("initial_prompt", C.POINTER(C.c_int)), ("n_initial_prompt", C.c_int),
                ("carry_context", C.c_int), ("beam_size", C.c_int)]


class AsrSegment(C.Structure):
    """crispy_asr_segment"""
    _fields_ = [("t0", C.c_float), ("t1", C.c_float), ("text", C.c_char_p)]


class AsrWindow(C.Structure):
    """crispy_asr_window"""
    _fields_ = [("seek", C.c_int), ("seek_advance", C.c_int), ("n_tokens", C.c_int), ("decoder", C.c_int),
                ("failed", C.c_int), ("no_speech", C.c_int), ("temperature", C.c_float), ("no_speech_prob", C.c_float),
                ("avg_logprob", C.c_float), ("entropy", C.c_float)]


class AsrResult(C.Structure):
    """crispy_asr_result"""
    _fields_ = [("text", C.c_char_p), ("tokens", C.POINTER(C.c_int)), ("n_tokens", C.c_int),
                ("language_token", C.c_int), ("n_segments", C.c_int), ("segments", C.POINTER(AsrSegment)),
                ("n_windows", C.c_int), ("windows", C.POINTER(AsrWindow))]


ERR_CANCELLED = -7
# crispy_asr_progress_fn: void (*)(size_t samples_done, size_t samples_total, void *user)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_size_t, C.c_size_t, C.c_void_p)


class CrispyError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"crispy_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libcrispy_hip.so (built by `__graft_entry__.build()` / `make -C crispy_amd/csrc`)."""
    global _lib
    if _lib is None:
        _lib = load_library(LIB_PATH)
    return _lib


def load_variant(name: str) -> C.CDLL:
    """Another build of the same library next to the default one: libcrispy_hip_<name>.so (`make variants`: poison = the
    checker build whose RNNoise kernels fill their LDS with NaNs first).  A separate handle with its own state."""
    return load_library(os.path.join(_HERE, f"libcrispy_hip_{name}.so"))


def load_library(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  crispy_amd has no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64.so.7.  If this library pulled in /opt/rocm's copy
    # first, a later `import torch` would load a second HIP runtime into the process and find no GPU.
    # Loading torch first makes both share one runtime (same SONAME), and torch tensors' device
    # pointers are then directly usable by the *_device entry points.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    f32p = C.POINTER(C.c_float)
    L.crispy_last_error.restype = C.c_char_p
    L.crispy_version.restype = C.c_char_p
    L.crispy_device_count.restype = C.c_int
    L.crispy_abi_version.restype = C.c_int
    if L.crispy_abi_version() != ABI_VERSION:      # include/crispy_hip.h: CRISPY_ABI_VERSION
        raise RuntimeError(f"{path}: ABI version {L.crispy_abi_version()}, this binding was written for {ABI_VERSION}")
    L.crispy_rn_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_rn_destroy.argtypes = [C.c_void_p]
    L.crispy_rn_weights_from_file.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.crispy_rn_create_from_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_selftest_exception_guard.argtypes = [C.c_int]
    L.crispy_asr_vocab_specials.argtypes = [C.c_int, C.c_void_p]
    L.crispy_asr_language_token.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    L.crispy_rn_destroy.restype = None
    L.crispy_rn_reset.argtypes = [C.c_void_p, C.c_int]
    L.crispy_rn_n_streams.argtypes = [C.c_void_p]
    L.crispy_rn_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.crispy_rn_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_void_p]
    L.crispy_rn_process_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.crispy_rn_process_s16_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.crispy_rn_synchronize.argtypes = [C.c_void_p]
    L.crispy_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.crispy_host_unregister.argtypes = [C.c_void_p]
    L.crispy_rn_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.crispy_rn_last_kernel_ms.argtypes = [C.c_void_p, f32p, f32p]
    L.crispy_rn_stage_tansig_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.crispy_rn_debug_capture.argtypes = [C.c_void_p, C.c_int]
    L.crispy_rn_debug_read.argtypes = [C.c_void_p, C.c_int, f32p, C.c_size_t]
    L.crispy_mel_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_mel_destroy.argtypes = [C.c_void_p]
    L.crispy_mel_destroy.restype = None
    L.crispy_mel_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    L.crispy_mel_compute_device.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
    L.crispy_mel_synchronize.argtypes = [C.c_void_p]
    L.crispy_asr_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_asr_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.crispy_asr_finalize.argtypes = [C.c_void_p]
    L.crispy_asr_free.argtypes = [C.c_void_p]
    L.crispy_asr_free.restype = None
    L.crispy_asr_hparams_get.argtypes = [C.c_void_p, C.c_void_p]
    L.crispy_asr_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    L.crispy_asr_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.crispy_asr_synchronize.argtypes = [C.c_void_p]
    L.crispy_asr_set_suppress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.crispy_asr_decode_greedy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
    L.crispy_asr_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_asr_load_resident.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_asr_memory_info.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.crispy_asr_token_text.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.crispy_asr_transcribe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
    L.crispy_asr_transcribe_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.crispy_asr_free_result.argtypes = [C.c_void_p]
    L.crispy_asr_transcribe_recording.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                                  PROGRESS_FN, C.c_void_p, C.POINTER(C.c_void_p)]
    L.crispy_asr_free_result.restype = None
    L.crispy_asr_set_precision.argtypes = [C.c_void_p, C.c_int]
    L.crispy_asr_stage_logits_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.crispy_asr_decode_timestamps_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                      C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                      C.c_void_p]
    L.crispy_asr_decode_window_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crispy_mel_window_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crispy_asr_decode_greedy_lang_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crispy_asr_detect_language_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.crispy_asr_transcribe_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.crispy_resampler_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.crispy_resampler_destroy.argtypes = [C.c_void_p]
    L.crispy_resampler_destroy.restype = None
    L.crispy_resampler_out_len.argtypes = [C.c_long]
    L.crispy_resampler_out_len.restype = C.c_long
    L.crispy_resampler_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_float,
                                                  C.c_int, C.c_void_p, C.c_long, C.c_void_p]
    L.crispy_resampler_synchronize.argtypes = [C.c_void_p]
    return L


def check(rc: int, L: "C.CDLL | None" = None) -> None:
    if rc != 0:
        raise CrispyError(rc, (L or lib()).crispy_last_error().decode("utf-8", "replace"))


class Specials(C.Structure):
    """`crispy_asr_specials` (include/crispy_hip.h): special token ids of a whisper.cpp vocabulary of n_vocab entries."""
    _fields_ = [(n, C.c_int) for n in ("eot", "sot", "lang0", "n_lang", "translate", "transcribe", "solm", "prev", "nosp",
                                       "notimestamps", "beg", "multilingual")]


def vocab_specials(n_vocab: int) -> Specials:
    sp = Specials()
    check(lib().crispy_asr_vocab_specials(int(n_vocab), C.byref(sp)))
    return sp
