"""Host-side mirror of the reference's RNNoise surface, backed by the HIP library.

* `DenoiseState`        -- nnnoiseless::DenoiseState for B streams at once
                           (`new`: src-tauri/src/audio.rs:229, `process_frame`: audio.rs:268).
* `RnnNoiseProcessor`   -- the adapter of audio.rs:202-315 (`push_sample`, x32768, clamp, volume,
                           first-frame drop, optional input LinearResampler), batched: one
                           processor object drives B streams that are pushed in lock step.
* `LinearResampler`     -- audio.rs:73-134.

All arithmetic of `process_frame` runs in libcrispy_hip.so on the GPU; nothing here falls back
to a CPU implementation."""
from __future__ import annotations

import ctypes as C
from collections import deque
from typing import Optional

import numpy as np

from . import _native as N

FRAME_SIZE = N.RN_FRAME  # nnnoiseless::FRAME_SIZE


class DenoiseState:
    """B independent `DenoiseState`s living in HBM.

    `process_frame(out, inp)` keeps the reference's argument order and returns the VAD
    probabilities; arrays are [B, 480] (or [480] when B == 1), f32 in int16 range."""

    def __init__(self, weights, n_streams: int = 1, device: int = 0, lib=None):
        """weights: the 87 503-byte int8 blob, or the path of an rnnoise-nu text model file
        (`crispy_rn_create_from_file`).  lib: another build of the library (`_native.load_variant`), tests only."""
        self._L = lib if lib is not None else N.lib()
        self._h = C.c_void_p()
        self.n_streams = int(n_streams)
        self.device = int(device)
        if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            import os
            N.check(self._L.crispy_rn_create_from_file(os.fsencode(weights), self.n_streams, self.device,
                                                       C.byref(self._h)), self._L)
            return
        w = np.ascontiguousarray(weights, dtype=np.int8)
        N.check(self._L.crispy_rn_create(w.ctypes.data_as(C.c_void_p), w.size, self.n_streams,
                                         self.device, C.byref(self._h)), self._L)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.crispy_rn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference-shaped single tick ------------------------------------------------------
    def process_frame(self, output: np.ndarray, input: np.ndarray):
        x = np.ascontiguousarray(input, dtype=np.float32)
        if x.size != self.n_streams * FRAME_SIZE:
            raise ValueError(f"process_frame: expected {self.n_streams}x{FRAME_SIZE} samples, got {x.size}")
        if output.dtype != np.float32 or not output.flags["C_CONTIGUOUS"] or output.size != x.size:
            raise ValueError("process_frame: output must be a contiguous float32 array of the input's size")
        vad = np.empty(self.n_streams, dtype=np.float32)
        N.check(self._L.crispy_rn_process(self._h, x.ctypes.data, output.ctypes.data, vad.ctypes.data,
                                          1, N.LAYOUT_TBF), self._L)
        return float(vad[0]) if self.n_streams == 1 else vad

    # -- batched host arrays ---------------------------------------------------------------
    def process(self, x: np.ndarray, layout: str = "tbf"):
        """x: [T, B, 480] ('tbf') or [B, T, 480] ('btf') -> (out like x, vad [T, B])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        lay = N.LAYOUT_TBF if layout == "tbf" else N.LAYOUT_BTF
        if x.ndim != 3 or x.shape[2] != FRAME_SIZE:
            raise ValueError("process: x must be [T,B,480] or [B,T,480]")
        T = x.shape[0] if layout == "tbf" else x.shape[1]
        Bn = x.shape[1] if layout == "tbf" else x.shape[0]
        if Bn != self.n_streams:
            raise ValueError(f"process: {Bn} streams given, handle has {self.n_streams}")
        out = np.empty_like(x)
        vad = np.empty((T, Bn), dtype=np.float32)
        N.check(self._L.crispy_rn_process(self._h, x.ctypes.data, out.ctypes.data, vad.ctypes.data, T, lay), self._L)
        return out, vad

    def process_s16(self, x: np.ndarray, layout: str = "tbf", out: np.ndarray | None = None, vad: np.ndarray | None = None):
        """`crispy_rn_process_s16`: int16 PCM [T, B, 480] / [B, T, 480] in -> (int16 out like x, vad [T, B]).  A sample s
        enters process_frame as float(s); the output is trunc(clamp(y / 32768, -1, 1) * 32767) -- the adapter's scaling and
        clamp (audio.rs:270-273) and the WAV writer's quantisation (recording.rs:109-110).  Half the PCIe bytes of `process`."""
        if x.dtype != np.int16 or not x.flags.c_contiguous:
            raise ValueError("process_s16: a C-contiguous int16 array is required")
        lay = N.LAYOUT_TBF if layout == "tbf" else N.LAYOUT_BTF
        if x.ndim != 3 or x.shape[2] != FRAME_SIZE:
            raise ValueError("process_s16: x must be [T,B,480] or [B,T,480]")
        T = x.shape[0] if layout == "tbf" else x.shape[1]
        Bn = x.shape[1] if layout == "tbf" else x.shape[0]
        if Bn != self.n_streams:
            raise ValueError(f"process_s16: {Bn} streams given, handle has {self.n_streams}")
        out = np.empty_like(x) if out is None else out
        vad = np.empty((T, Bn), dtype=np.float32) if vad is None else vad
        if out.dtype != np.int16 or out.shape != x.shape or not out.flags.c_contiguous or vad.shape != (T, Bn):
            raise ValueError("process_s16: out / vad shape or type mismatch")
        N.check(self._L.crispy_rn_process_s16(self._h, x.ctypes.data, out.ctypes.data, vad.ctypes.data, T, lay), self._L)
        return out, vad

    def process_s16_device(self, d_in: int, d_out: int, n_frames: int, d_vad: int = 0, layout: str = "tbf", stream: int = 0):
        lay = N.LAYOUT_TBF if layout == "tbf" else N.LAYOUT_BTF
        N.check(self._L.crispy_rn_process_s16_device(self._h, d_in, d_out, d_vad or None, int(n_frames), lay, stream or None), self._L)

    @staticmethod
    def register_host(arr: np.ndarray):
        """Page-lock a host array the caller reuses across `process` calls (crispy_host_register): copies become DMA."""
        N.check(N.lib().crispy_host_register(arr.ctypes.data, arr.nbytes))

    @staticmethod
    def unregister_host(arr: np.ndarray):
        N.check(N.lib().crispy_host_unregister(arr.ctypes.data))

    def process_into(self, x: np.ndarray, out: np.ndarray, vad: np.ndarray, layout: str = "tbf"):
        """`process` into caller-owned (e.g. registered) arrays: x, out [T,B,480] / [B,T,480] float32 contiguous,
        vad [T,B]."""
        lay = N.LAYOUT_TBF if layout == "tbf" else N.LAYOUT_BTF
        T = x.shape[0] if layout == "tbf" else x.shape[1]
        Bn = x.shape[1] if layout == "tbf" else x.shape[0]
        if x.dtype != np.float32 or out.dtype != np.float32 or not x.flags.c_contiguous or not out.flags.c_contiguous:
            raise ValueError("process_into: float32 C-contiguous arrays required")
        if Bn != self.n_streams or out.shape != x.shape or vad.shape != (T, Bn):
            raise ValueError("process_into: shape mismatch")
        N.check(self._L.crispy_rn_process(self._h, x.ctypes.data, out.ctypes.data, vad.ctypes.data, T, lay), self._L)

    # -- device-resident tensors (torch is only the allocator here) --------------------------
    def process_device(self, d_in: int, d_out: int, n_frames: int, d_vad: int = 0, d_taps: int = 0,
                       layout: str = "tbf", stream: int = 0):
        lay = N.LAYOUT_TBF if layout == "tbf" else N.LAYOUT_BTF
        N.check(self._L.crispy_rn_process_device(self._h, d_in, d_out, d_vad or None, d_taps or None,
                                                 int(n_frames), lay, stream or None), self._L)

    def stage_tansig_device(self, d_x: int, d_y: int, n: int, sigmoid: bool = False):
        """tansig_approx / sigmoid_approx as the frame kernel evaluates them (stage entry point for parity tests)."""
        N.check(self._L.crispy_rn_stage_tansig_device(self._h, d_x, d_y, int(n), int(sigmoid), None), self._L)

    def synchronize(self):
        N.check(self._L.crispy_rn_synchronize(self._h), self._L)

    def reset(self, stream: int = -1):
        N.check(self._L.crispy_rn_reset(self._h, int(stream)), self._L)

    def set_timing(self, enable: bool):
        N.check(self._L.crispy_rn_set_timing(self._h, int(enable)), self._L)

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        N.check(self._L.crispy_rn_last_kernel_ms(self._h, C.byref(a), C.byref(b)), self._L)
        return a.value, b.value

    def debug_capture(self, enable: bool):
        N.check(self._L.crispy_rn_debug_capture(self._h, int(enable)), self._L)

    def debug_read(self, stream: int) -> np.ndarray:
        d = np.empty(N.RN_DBG_FLOATS, dtype=np.float32)
        N.check(self._L.crispy_rn_debug_read(self._h, int(stream), d.ctypes.data_as(C.POINTER(C.c_float)), d.size), self._L)
        return d


class LinearResampler:
    """Streaming 2-tap linear interpolation resampler (audio.rs:73-134)."""

    def __init__(self, input_rate: float, output_rate: float):
        self.input_rate = np.float32(input_rate)
        self.output_rate = np.float32(output_rate)
        self.last_sample = np.float32(0.0)
        self.has_last = False
        self.input_pos = 0.0
        self.next_output_pos = 0.0

    def rates(self):
        return float(self.input_rate), float(self.output_rate)

    def set_rates(self, input_rate: float, output_rate: float):
        self.__init__(input_rate, output_rate)

    def process_sample(self, sample, emit):
        sample = np.float32(sample)
        if abs(self.input_rate - self.output_rate) < 1.0:
            emit(sample)
            return
        if not self.has_last:
            self.last_sample = sample
            self.has_last = True
            self.input_pos = 0.0
            self.next_output_pos = 0.0
            return
        self.input_pos += 1.0
        step = float(np.float32(self.input_rate / self.output_rate))
        while self.next_output_pos <= self.input_pos:
            t = np.float32(self.next_output_pos - (self.input_pos - 1.0))
            t = np.float32(min(max(t, np.float32(0.0)), np.float32(1.0)))
            emit(np.float32(self.last_sample + (sample - self.last_sample) * t))
            self.next_output_pos += step
        self.last_sample = sample


class RnnNoiseProcessor:
    """audio.rs:202-315 for B lock-stepped streams: `push_sample(samples[B])` returns None or an
    array [n, B] of denoised samples (n = 480 per completed frame), scaled and clamped exactly
    as the reference does (x32768 in, /32768 + clamp(-1, 1) * volume out, first frame dropped)."""

    def __init__(self, weights: np.ndarray, input_rate: float, output_rate: float, volume: float,
                 n_streams: int = 1, device: int = 0):
        if abs(input_rate - 48000.0) >= 1.0:
            self.input_rate = 48000.0
            self.input_resamplers: Optional[list] = [LinearResampler(input_rate, 48000.0) for _ in range(n_streams)]
        else:
            self.input_rate = float(input_rate)
            self.input_resamplers = None
        self.output_rate = float(output_rate)
        self.volume = float(min(max(volume, 0.0), 1.0))
        self.n_streams = n_streams
        self.first_frame = True
        self.max_output_len = int(self.input_rate)
        self.denoise = DenoiseState(weights, n_streams, device)
        self.input_buf: deque = deque()
        self.output_buf: deque = deque()
        self.resample_pos = 0.0

    def set_volume(self, volume: float):
        self.volume = float(min(max(volume, 0.0), 1.0))

    def produced_rate_hz(self) -> float:
        return self.input_rate

    def push_sample(self, samples) -> Optional[np.ndarray]:
        samples = np.atleast_1d(np.asarray(samples, dtype=np.float32))
        if samples.size != self.n_streams:
            raise ValueError("push_sample: one sample per stream")
        rows = []
        if self.input_resamplers is not None:
            per_stream = [[] for _ in range(self.n_streams)]
            for b, rs in enumerate(self.input_resamplers):
                rs.process_sample(samples[b], per_stream[b].append)
            for k in range(len(per_stream[0])):  # lock-stepped streams emit equal counts
                rows.append(np.array([per_stream[b][k] for b in range(self.n_streams)], dtype=np.float32))
        else:
            rows.append(samples)
        acc = []
        for row in rows:
            if len(self.input_buf) >= self.max_output_len:
                self.input_buf.popleft()
            self.input_buf.append(row)
            if len(self.input_buf) >= FRAME_SIZE:
                frame = np.stack([self.input_buf.popleft() for _ in range(FRAME_SIZE)], axis=1)  # [B,480]
                frame = np.ascontiguousarray(frame * np.float32(32768.0))
                out = np.empty_like(frame)
                self.denoise.process_frame(out, frame)
                out = np.clip(out / np.float32(32768.0), -1.0, 1.0).astype(np.float32) * np.float32(self.volume)
                if self.first_frame:
                    self.first_frame = False
                    continue
                for i in range(FRAME_SIZE):
                    if len(self.output_buf) >= self.max_output_len:
                        self.output_buf.popleft()
                    self.output_buf.append(out[:, i])
                acc.append(out.T)
        if not acc:
            return None
        return np.concatenate(acc, axis=0)

    def next_sample(self) -> np.ndarray:
        zero = np.zeros(self.n_streams, dtype=np.float32)
        if len(self.output_buf) < 2:
            return zero
        step = self.input_rate / self.output_rate
        while self.resample_pos >= 1.0:
            self.output_buf.popleft()
            self.resample_pos -= 1.0
            if len(self.output_buf) < 2:
                return zero
        s0, s1 = self.output_buf[0], self.output_buf[1]
        frac = np.float32(self.resample_pos)
        self.resample_pos += step
        return (s0 + (s1 - s0) * frac).astype(np.float32)


REC_SAMPLE_RATE = 48000          # recording::SAMPLE_RATE (recording.rs:8)


class CaptureBuffers:
    """`push_mono_to_buffers` (audio.rs:682-730), the per-sample glue of the capture callback: feed the mono sample
    to the noise suppressor (or pass it through when none is active), resample whatever it produced to the 48 kHz
    recording rate with the caller's `LinearResampler` -- re-configured only when a rate changed by >= 1 Hz --,
    append to the recording ring (10 s cap, oldest dropped), and accumulate the level meter's sum / count
    (`rms()` is the `(sum / frames).sqrt()` of audio.rs:780)."""

    def __init__(self):
        self.rec_resampler = LinearResampler(float(REC_SAMPLE_RATE), float(REC_SAMPLE_RATE))
        self.rec_buffer = deque()
        self.max_len = REC_SAMPLE_RATE * 10
        self.sum = np.float32(0.0)
        self.frames = np.float32(0.0)

    def push_mono(self, mono: float, ns: "Optional[RnnNoiseProcessor]", raw_input_rate_hz: float) -> None:
        mono = np.float32(mono)
        if ns is not None:
            produced_rate = ns.produced_rate_hz()
            out = ns.push_sample([mono])
            samples = None if out is None else out[:, 0]
        else:
            produced_rate = float(raw_input_rate_hz)
            samples = np.array([mono], dtype=np.float32)
        if samples is not None:
            cur_in, cur_out = self.rec_resampler.rates()
            if abs(cur_in - produced_rate) >= 1.0 or abs(cur_out - REC_SAMPLE_RATE) >= 1.0:
                self.rec_resampler.set_rates(produced_rate, float(REC_SAMPLE_RATE))
            emitted = []
            for s in samples:
                self.rec_resampler.process_sample(s, emitted.append)
            for o in emitted:
                if len(self.rec_buffer) >= self.max_len:
                    self.rec_buffer.popleft()
                self.rec_buffer.append(np.float32(o))
        self.sum = np.float32(self.sum + mono * mono)
        self.frames = np.float32(self.frames + np.float32(1.0))

    def rms(self) -> float:
        return float(np.sqrt(self.sum / self.frames)) if self.frames > 0 else 0.0

    def reset_level(self) -> None:
        self.sum = np.float32(0.0)
        self.frames = np.float32(0.0)
