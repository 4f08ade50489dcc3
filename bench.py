#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched RNNoise (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (high-pass -> frame kernel -> history roll, i.e. one
`crispy_rn_process_device` call) over one batch of synthetic 48 kHz audio that is already
resident in HBM: `--streams` concurrent streams (default 4096) x `--frames` consecutive 10 ms
frames (default 100 = 1 s of audio per stream).  Metric: concurrent real-time 48 kHz streams per
GPU = frames/s / 100, summed over ranks (streams shard with no data-path collective).

The JSON line also carries
  roofline      HBM roofline of the dominant kernel (rn_frame_kernel): algorithmic bytes per launch
                (3840 B per stream-frame) / its average duration measured with hipEvents on the
                launch stream, against the 8 TB/s HBM3E peak.
  cpu_baseline  the C oracle (CPU restatement of the reference algorithm; the reference's Rust crates
                cannot be built here) timed on this box's host cores on a bounded sample, rank 0, N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_STREAM_FRAME = 3840  # 480 f32 in + 480 f32 out
FLOPS_PER_STREAM_FRAME = 0.45e6  # gain network 175 k + three 960-point transforms 75 k + pitch 120 k + bands / DCT / filters 80 k
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector, 256 CUs x 256 flop / clock x 2.4 GHz


class _ClockSampler:
    """Shader clock of the GPU while a loop runs, read from sysfs by a side thread every 0.2 s (never from inside the
    loop): /sys/class/drm/card*/device/pp_dpm_sclk marks the current level with '*'.  None where the file is missing."""

    def __init__(self, local_rank: int = 0):
        import glob
        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.path = None
        # the DRM card of THIS HIP device, by PCI address (a box shows every GPU of its host in sysfs, the job owns one)
        try:
            import torch
            pr = torch.cuda.get_device_properties(local_rank)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for q in self.paths:
                if want in os.path.realpath(os.path.dirname(q)):
                    self.path = q
        except Exception:
            pass
        self.by_pci = self.path is not None
        self.samples = []
        self._stop = False
        self._th = None

    def _read_one(self, path):
        import re
        try:
            with open(path) as f:
                for ln in f:
                    if "*" in ln:
                        m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
                        if m:
                            return int(m.group(1))
        except OSError:
            pass
        return None

    def _read(self):
        if self.path:
            return self._read_one(self.path)
        # no PCI match: the card under load is the one with the highest clock
        vals = [v for v in (self._read_one(q) for q in self.paths) if v]
        return max(vals) if vals else None

    def __enter__(self):
        import threading
        if self.paths:
            def loop():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.2)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def mean(self):
        return sum(self.samples) / len(self.samples) if self.samples else None


def _host_threads() -> int:
    """Host threads this process can actually run at once: the affinity mask, cut down to the cgroup's CPU quota (the GPU
    box hands a 1-GPU job ~16 CPUs' worth of a 256-thread host: 256 runnable threads on that quota measured SLOWER than
    32 -- r03b: 763 against 1 205 streams)."""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = max(1, os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n


def cpu_baseline(frames_per_thread: int = 2500, reps: int = 12):
    """Oracle (kind 'port') on host cores, BASELINE.md section 3: (a) ONE thread, one stream -- how the reference uses
    nnnoiseless (one stream on the audio thread, audio.rs:751-787) -- and (b) every host thread this process may use,
    one independent stream per thread.  `value` is (b); (a) rides along as `single_thread`.  Bounded: each leg is
    `frames_per_thread x reps` frames per thread (~10 s)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    from crispy_amd import synthetic_weights, synth_audio
    from tests import oracle_lib as O

    # prefer a -march=native build of the same source for the baseline; fall back to the generic one
    lib_path = None
    try:
        out = os.path.join(ROOT, "gpurun_out", "liboracle_native.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-ffp-contract=off", "-shared",
                        "-o", out, os.path.join(ROOT, "oracle", "rnnoise_oracle.c"), "-lm"],
                       check=True, capture_output=True)
        lib_path = out
    except Exception:
        pass
    if lib_path:
        import ctypes as C
        L = C.CDLL(lib_path)
        f32p = C.POINTER(C.c_float)
        L.rno_create.restype = C.c_void_p
        L.rno_create.argtypes = [C.c_void_p, C.c_size_t]
        L.rno_process_frames.argtypes = [C.c_void_p, f32p, f32p, C.c_int, f32p]
        L.rno_destroy.argtypes = [C.c_void_p]
    else:
        L = O.lib()
    w = synthetic_weights(0)
    build = "-O3 -march=native" if lib_path else "-O2"

    def leg(n_threads, reps=reps):
        xs = [np.ascontiguousarray(synth_audio.stream_np(b % 64, frames_per_thread, silent=False) * np.float32(32768.0))
              for b in range(n_threads)]
        handles = [L.rno_create(w.ctypes.data, w.size) for _ in range(n_threads)]
        outs = [np.empty_like(x) for x in xs]

        def run(i):        # the same buffer again and again (the state carries on)
            for _ in range(reps):
                L.rno_process_frames(handles[i], O.fp(outs[i]), O.fp(xs[i]), frames_per_thread, None)

        with ThreadPoolExecutor(n_threads) as ex:
            list(ex.map(lambda i: L.rno_process_frames(handles[i], O.fp(outs[i]), O.fp(xs[i]), frames_per_thread, None),
                        range(n_threads)))  # warm
            t0 = time.perf_counter()
            list(ex.map(run, range(n_threads)))
            dt = time.perf_counter() - t0
        for h in handles:
            L.rno_destroy(h)
        fps = n_threads * frames_per_thread * reps / dt
        return {"value": fps / 100.0, "unit": "concurrent real-time 48 kHz streams", "cores": n_threads, "kind": "port",
                "sample": f"{n_threads} stream(s) x {frames_per_thread * reps} frames (tone+noise), C oracle {build}, "
                          f"one stream per thread, {dt:.1f} s"}

    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    one = leg(1)
    n = _host_threads()
    probe = None
    if n > 32:
        # No cgroup quota visible but a large affinity mask: a container share can still be much smaller than the mask.
        # One short pass with every thread tells how many cores' worth of work actually runs at once; the timed leg then
        # uses that many threads (oversubscribed threads on a throttled share measured slower than fewer threads).
        probe = leg(n, reps=1)
        eff = max(1, int(round(probe["value"] / max(one["value"], 1e-9))))
        if eff < 0.6 * n:
            n = eff
    many = leg(n)
    many["single_thread"] = one
    many["host"] = {"cpu_model": cpu_model, "os_cpu_count": os.cpu_count(), "affinity_threads": _host_threads(),
                    "threads_used": n, "probe_with_all_threads": None if probe is None else probe["value"]}
    return many


def committed_pmc():
    """The newest committed counter summary of rn_frame_kernel: (file, streams, per-stream-frame counters).  One reader
    for both layouts under profiles/: what tools/collect_profiles.sh writes (`kernels.<name>.per_stream_frame`) and what
    tools/summarize_pmc.py makes of it (`rn_frame_kernel.per_stream_frame` + `config`)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc.json")), reverse=True):
        try:
            pm = json.load(open(f))
            if "kernels" in pm:
                psf, streams = pm["kernels"]["rn_frame_kernel"]["per_stream_frame"], pm["streams"]
            else:
                psf, streams = pm["rn_frame_kernel"]["per_stream_frame"], pm["config"]["streams"]
            if all(k in psf for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU")):
                return os.path.relpath(f, ROOT), int(streams), psf
        except Exception:
            continue
    raise FileNotFoundError("no counter summary under profiles/")


def measure_traffic(B: int, T: int):
    """HBM-side traffic of rn_frame_kernel measured on THIS box for THIS build: two separate `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no trace domains beside them) over a child process that
    runs three steps of the bench shape (`bench.py --pmc-child`).  Corrections as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950 (x2), WRITE_SIZE is exact; both in KiB.
    Returns bytes per stream-frame, or raises."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    steps = 3
    kib = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"crispy_pmc_{ctr}_", dir="/tmp")
        try:
            r = subprocess.run([rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.abspath(__file__), "--pmc-child", "--streams", str(B), "--frames", str(T),
                                "--steps", str(steps)], capture_output=True, text=True, timeout=180, env=env, cwd="/tmp")
            if r.returncode != 0:
                raise RuntimeError(f"rocprofv3 --pmc {ctr} exited with {r.returncode}: {r.stderr[-300:]}")
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "rn_frame_kernel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        tot += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                raise RuntimeError(f"no rn_frame_kernel dispatch in the {ctr} pass")
            kib[ctr] = tot
        finally:
            shutil.rmtree(d, ignore_errors=True)
    sf = float(B) * T * steps
    return {"fetch_bytes": 2.0 * kib["FETCH_SIZE"] * 1024.0 / sf, "write_bytes": kib["WRITE_SIZE"] * 1024.0 / sf}


def pmc_child(args):
    """What `measure_traffic` profiles: `--steps` steps of the bench shape, nothing else, no output."""
    import torch
    from crispy_amd import synthetic_weights, synth_audio
    from crispy_amd.denoise import DenoiseState
    dev = torch.device("cuda", 0)
    ds = DenoiseState(synthetic_weights(0), args.streams, 0)
    d_in = synth_audio.batch_torch(args.streams, args.frames, dev, first_stream=0, seed=0)
    d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()
    for _ in range(args.steps):
        ds.process_device(d_in.data_ptr(), d_out.data_ptr(), args.frames)
    ds.synchronize()


def encoder_flops(hp) -> float:
    """Encoder FLOPs per 30 s clip from the model's dimensions (SURVEY.md 8d's 36.9 GFLOP for tiny, 87.4 for base)."""
    d, L, T = hp.n_audio_state, hp.n_audio_layer, hp.n_audio_ctx
    conv = 2.0 * (2 * T) * d * 3 * hp.n_mels + 2.0 * T * d * 3 * d
    block = 4 * 2.0 * T * d * d + 2 * 2.0 * T * T * d + 2 * 2.0 * T * d * 4 * d
    return conv + L * block


def asr_leg(local_rank: int, clips: int = 64, new_tokens: int = 32, ggml_path: str | None = None):
    """Second half of the headline metric: Whisper-tiny RTFx on one GPU (BASELINE configs[2]/[3]):
    `clips` x 30 s of 16 kHz audio resident in HBM -> log-mel -> encoder -> greedy decode of
    `new_tokens` tokens (random-init weights never emit EOT, so the decode length is fixed).
    `ggml_path`: a whisper.cpp model file supplied at run time (SURVEY.md 8d cfg 3: "if a ggml-tiny.bin is supplied at
    run time use it") replaces the seeded random-init Whisper-tiny; the audio stays synthetic."""
    import numpy as np
    import torch

    from crispy_amd.asr import LogMel, WhisperEngine, WhisperModel
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

    if ggml_path:
        model = WhisperEngine(ggml_path, device=local_rank)
        hp = model.hp
        model_label = f"GGML model file {os.path.basename(ggml_path)} (d {hp.n_audio_state}, {hp.n_audio_layer}+{hp.n_text_layer} layers)"
    else:
        hp = HParams.tiny()
        model = WhisperModel(hp, synthetic_whisper_weights(hp, 0), device=local_rank)
        model_label = "whisper-tiny architecture, seeded random-init weights"
    lm = LogMel(hp.n_mels, device=local_rank)
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev).manual_seed(0)
    pcm = torch.randn(clips, 480000, generator=g, device=dev) * 0.1
    melt = torch.zeros(clips, 3002, hp.n_mels, device=dev)
    enc = torch.empty(clips, 1500, hp.n_audio_state, device=dev)
    lens = np.full(clips, 480000)
    from crispy_amd import _native as N
    sp = N.vocab_specials(hp.n_vocab)
    prompt = ([sp.sot, sp.sot + 1, sp.transcribe, sp.notimestamps] if hp.n_vocab >= 51865 else [sp.sot, sp.notimestamps])
    torch.cuda.synchronize()

    def mel():
        lm.compute_device(pcm.data_ptr(), 480000, lens, 0, melt.data_ptr())
        lm.synchronize()

    def encode():
        model.encode_device(melt.data_ptr(), clips, enc.data_ptr())
        model.synchronize()

    def decode():
        model.decode_greedy_device(enc.data_ptr(), clips, prompt, new_tokens)

    def measure():
        t = {}
        for name, fn, reps in (("logmel", mel, 5), ("encoder", encode, 3), ("decode", decode, 1)):
            fn()  # warm-up
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t[name] = (time.perf_counter() - t0) / reps
        return t

    def decode_n(n):
        model.decode_greedy_device(enc.data_ptr(), clips, prompt, n)       # the captured step for this mode and reach is warm
        t0 = time.perf_counter()
        model.decode_greedy_device(enc.data_ptr(), clips, prompt, n)
        return time.perf_counter() - t0

    def single_clip():
        """The literal drop-in call: ONE 30 s chunk from host memory through the product path (PCM upload, log-mel,
        encoder, cross K|V, prompt, 32 greedy tokens, ids back), as managers/transcription.rs:183-185 makes it."""
        x1 = pcm[0].cpu().numpy()
        model.transcribe_tokens([x1], prompt, new_tokens)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            model.transcribe_tokens([x1], prompt, new_tokens)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def default_call():
        """`engine.transcribe(&audio, &TranscribeOptions::default())` (managers/transcription.rs:183-185) as it is: opts =
        NULL -- language detected, timestamp tokens, whisper_full's seek loop with previous-text conditioning -- on one 28 s
        clip from host memory.  Random-init weights decide how many windows and tokens that is; both are reported."""
        import ctypes as C
        from crispy_amd import _native as N
        from crispy_amd.asr import make_opts
        x1 = np.ascontiguousarray(pcm[0].cpu().numpy()[:16000 * 28])

        def call(opts):
            res = C.c_void_p()
            N.check(N.lib().crispy_asr_transcribe(model._h, x1.ctypes.data, x1.size, C.byref(opts) if opts is not None else None,
                                                  C.byref(res)))
            r = C.cast(res, C.POINTER(N.AsrResult)).contents
            temps = [float(r.windows[i].temperature) for i in range(r.n_windows)]
            out = (int(r.n_tokens), int(r.n_segments), int(r.n_windows), sum(t > 0 for t in temps),
                   sum(int(r.windows[i].no_speech) for i in range(r.n_windows)))
            N.lib().crispy_asr_free_result(res)
            return out

        def timed(opts):
            n_tok, n_seg, n_win, n_fb, n_ns = call(opts)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                call(opts)
                ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            return {"ms": t * 1e3, "tokens": n_tok, "segments": n_seg, "windows": n_win, "windows_re_decoded": n_fb,
                    "windows_dropped_as_silence": n_ns, "ms_per_token": t * 1e3 / max(n_tok, 1), "rtfx": 28.0 / t}

        # opts = NULL walks whisper_full's temperature ladder wherever a window fails its thresholds -- on random-init
        # weights (log-probability ~ -10 per token, no end of text) that is every window, all the way to 1.0 with five
        # sampling decoders per pass, exactly as whisper.cpp would on such logits; the single greedy pass per window
        # (temperature_inc < 0) is timed beside it: what a window that decodes well costs
        out = timed(None)
        out["what"] = ("crispy_asr_transcribe(h, pcm, n, opts = NULL): language detection, timestamp rules, seek loop, previous-text "
                       "conditioning, no-speech rule, temperature fallback; 28 s of audio from host memory")
        out["one_greedy_pass_per_window"] = timed(make_opts(timestamps=True, fallback=False))
        # the same call for 64 clips at once, every window walking the whole ladder (random-init logits): since round 5 the
        # fallback passes of ALL failed clips decode side by side, best_of rows per clip over one cross K|V (they used to run
        # clip after clip: ~64 x the single-clip time by construction)
        from crispy_amd.asr import transcribe_batch
        clips = [np.ascontiguousarray(pcm[i % pcm.shape[0]].cpu().numpy()[:16000 * 28]) for i in range(64)]
        transcribe_batch(model, clips, timestamps=True)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            res = transcribe_batch(model, clips, timestamps=True, with_segments=True)
            ts.append(time.perf_counter() - t0)
        out["batch_ladder"] = {"clips": 64, "ms": float(np.median(ts)) * 1e3, "single_clip_ms": out["ms"],
                               "ratio_to_single_clip": float(np.median(ts)) * 1e3 / out["ms"],
                               "windows": int(sum(len(r[4]) for r in res)),
                               "windows_re_decoded": int(sum(sum(w["temperature"] > 0 for w in r[4]) for r in res)),
                               "rtfx": 64 * 28.0 / float(np.median(ts))}
        return out

    times = measure()                      # default precision: f32 operands, the mode the oracle parity is pinned in
    one = single_clip()
    longer = {n: decode_n(n) for n in (64, 224)}      # SURVEY cfg 4 asks for 64 tokens; a full 30 s window can take 224
    model.set_precision(1)                 # the reference's precision: f16 operands / f32 accumulation (ggml numerics)
    times16 = measure()
    one16 = single_clip()
    longer16 = {n: decode_n(n) for n in (64, 224)}
    try:
        default16 = default_call()
    except Exception as e:                 # (a supplied model file without timestamp tokens, ...)
        default16 = {"error": str(e)[:200]}
    model.set_precision(0)
    # CPU beside it (kind "port"): the numpy restatement in SINGLE precision (OpenBLAS sgemm on the host threads this
    # process may use -- what a CPU engine's matrix products amount to) on ONE 30 s clip: C log-mel + encoder + greedy
    # steps of the KV-cached decoder.  The float64 run of the same code is the parity oracle and is not what is timed.
    cpu = None
    try:
        if ggml_path:
            raise RuntimeError("no CPU restatement of a supplied model file (its tensors live in the library)")
        from crispy_amd import synth_audio
        from crispy_amd.mel_filters import whisper_mel_filters
        from oracle import whisper_oracle as WO
        from tests import oracle_lib as O
        W = synthetic_whisper_weights(hp, 0)
        x1 = synth_audio.clip16k_np(0, 480000)
        WO.encoder_forward(W, hp, O.oracle_logmel(x1[:32000], whisper_mel_filters(hp.n_mels)), upto_layer=1, dtype=np.float32)  # warm BLAS
        t0 = time.perf_counter()
        e1 = WO.encoder_forward(W, hp, O.oracle_logmel(x1, whisper_mel_filters(hp.n_mels)), dtype=np.float32)
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        dc = WO.DecoderCache(W, hp, e1, dtype=np.float32)
        for t in prompt[:-1]:
            dc.step(t)
        t_pre = time.perf_counter() - t0
        t0 = time.perf_counter()
        tok = prompt[-1]
        for _ in range(8):
            tok = int(np.argmax(dc.step(tok)))
        t_tok = (time.perf_counter() - t0) / 8
        cpu = {"value": 30.0 / (t_enc + t_pre + new_tokens * t_tok), "unit": "x real time (end to end, same token count)",
               "cores": _host_threads(), "kind": "port",
               "sample": f"1 clip of 30 s: C log-mel + float32 numpy/BLAS encoder {t_enc:.2f} s, cross K|V + prompt {t_pre:.2f} s, "
                         f"KV-cached decoder {t_tok * 1e3:.1f} ms/token x {new_tokens} tokens; single stream, as the reference "
                         f"runs one engine behind a mutex (managers/transcription.rs:27,178)"}
    except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU numbers
        cpu = {"error": str(e)}
    audio_s = clips * 30.0
    enc_flops = clips * encoder_flops(hp)          # 36.9 GFLOP per 30 s clip for Whisper-tiny
    # mode-1 bytes (NOTEBOOK 8.1): per layer LayerNorm x 2 (f32 in, f16 out), q|k|v, attention, out-projection and fc2
    # (f16 A, f32 residual in and out), fc1 (f16 in, 4 d f16 out) = 64 M d bytes with M = 1500 rows per clip; stem: the f16
    # log-mel, conv1's f16 output written and read, conv2's f32 output
    m_rows, d_enc = clips * hp.n_audio_ctx, hp.n_audio_state
    enc_bytes16 = hp.n_audio_layer * 64.0 * m_rows * d_enc + clips * 3000 * hp.n_mels * 2 + 2 * (2 * m_rows) * d_enc * 2 + m_rows * d_enc * 4
    steps16 = new_tokens + 4                       # positions of the decode call: the 4-token prompt + the new tokens
    dec_bytes16 = steps16 * (hp.n_text_layer * clips * hp.n_audio_ctx * 2 * hp.n_text_state * 2 + hp.n_vocab * hp.n_text_state * 2)
    total = sum(times.values())
    return {
        "model": model_label,
        "modes": "top-level figures of this block: precision mode 0 (exact f32 products on the f32-input matrix cores, the "
                 "mode the 1e-4 oracle parity is stated in); `f16_operand_mode`: precision mode 1 = the reference engine's "
                 "arithmetic (f16 operands, f32 accumulation), the Rust binding's default",
        "clips": clips, "audio_seconds": audio_s, "new_tokens": new_tokens,
        "logmel_ms": times["logmel"] * 1e3, "encoder_ms": times["encoder"] * 1e3,
        "decode_ms": times["decode"] * 1e3, "decode_ms_per_token": times["decode"] * 1e3 / new_tokens,
        "longer_decodes": {str(n): {"decode_ms": longer[n] * 1e3,
                                    "rtfx_end_to_end": audio_s / (times["logmel"] + times["encoder"] + longer[n]),
                                    "f16_operand_mode_rtfx_end_to_end": audio_s / (times16["logmel"] + times16["encoder"] + longer16[n])}
                           for n in (64, 224)},
        "f16_operand_mode": {"note": "crispy_asr_set_precision(1) = the reference's precision (whisper.cpp: f16 operands, f32 "
                                     "accumulation): encoder GEMMs + attention on v_mfma_f32_32x32x16_f16, activations that only "
                                     "feed a matrix product stored as f16, f16 cross K|V; checked against the f16-operand oracle "
                                     "(tests/test_gpu_whisper.py::test_f16_operand_mode_matches_the_f16_operand_oracle)",
                             "encoder_ms": times16["encoder"] * 1e3, "decode_ms": times16["decode"] * 1e3,
                             "rtfx_end_to_end": audio_s / sum(times16.values()),
                             "encoder_tflops": enc_flops / times16["encoder"] / 1e12,
                             "encoder_roofline": {"bound": "mfma", "achieved": enc_flops / times16["encoder"] / 1e12,
                                                  "peak": 2500.0, "unit": "TFLOP/s",
                                                  "frac": enc_flops / times16["encoder"] / 1e12 / 2500.0,
                                                  "note": "peak = dense f16 MFMA (v_mfma_f32_32x32x16_f16)"},
                             # the same pass against the OTHER roof: with K = d and an f32 residual stream every GEMM of
                             # this encoder sits left of the 312 flop / byte ridge (NOTEBOOK 8.1), so bytes are the price
                             "encoder_hbm_roofline": {"bound": "hbm", "achieved": enc_bytes16 / times16["encoder"] / 1e9,
                                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "frac": enc_bytes16 / times16["encoder"] / 1e9 / HBM_PEAK_GBS,
                                                      "bytes_per_pass": enc_bytes16,
                                                      "note": "algorithmic bytes of the tensors each kernel reads and writes "
                                                              "once (64 M d per layer + the convolution stem); L2 / "
                                                              "Infinity-Cache hits between producer and consumer not deducted"},
                             "decode_hbm_roofline": {"bound": "hbm", "achieved": dec_bytes16 / times16["decode"] / 1e9,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": dec_bytes16 / times16["decode"] / 1e9 / HBM_PEAK_GBS,
                                                     "note": "f16 cross K|V of all layers + the f16 token embedding, once per "
                                                             "decoded position"}},
        "single_clip": {"what": "ONE 30 s chunk, host PCM in, token ids out (crispy_asr_transcribe_tokens): the call the "
                                "reference makes per chunk (managers/transcription.rs:183-185)",
                        "ms": one * 1e3, "rtfx": 30.0 / one, "f16_operand_mode_ms": one16 * 1e3, "f16_operand_mode_rtfx": 30.0 / one16,
                        "default_options_f16_operand_mode": default16},
        "rtfx_logmel_encoder": audio_s / (times["logmel"] + times["encoder"]),
        "rtfx_end_to_end": audio_s / total,
        "logmel_roofline": {"bound": "hbm", "achieved": clips * 2.88e6 / times["logmel"] / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": clips * 2.88e6 / times["logmel"] / 1e9 / HBM_PEAK_GBS},
        "encoder_roofline": {"bound": "mfma", "achieved": enc_flops / times["encoder"] / 1e12, "peak": 157.3,
                             "unit": "TFLOP/s", "frac": enc_flops / times["encoder"] / 1e12 / 157.3,
                             "note": "peak = dense f32-input MFMA (v_mfma_f32_32x32x2_f32)"},
        "cpu_baseline": cpu,
    }


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def _init_dist(backend: str, local_rank: int, world: int):
    """torch.distributed over RCCL ("nccl") on the GPU box, gloo in the launcher's CPU dry run.  Returns
    (dist module or None, world size AS THE BACKEND REPORTS IT)."""
    # CRISPY_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (what can be tested on a 1-GPU box)
    if world <= 1 and os.environ.get("CRISPY_BENCH_FORCE_DIST") != "1":
        return None, 1
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist, dist.get_world_size()


def _gather_ms(dist, ms: float, device=None):
    """Per-rank elapsed milliseconds, rank order (one tiny all-gather; [ms] without a process group)."""
    if dist is None:
        return [ms]
    import torch
    t = torch.tensor([ms], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def _flush_c_stdout():
    # RCCL prints its version banner to C stdout when the communicator comes up; push it out before the timed
    # region on every rank so that rank 0's JSON line is the last thing on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)


def dry_run(args):
    """Launcher self-test (tests/test_sharding_gloo.py): the rank plumbing of a --gpus N job on CPU with gloo -- no
    HIP library, no GPU, a "step" is a 5 ms sleep.  The line it prints is labelled as such and carries no value."""
    rank, local_rank, world = _dist_env()
    fail = os.environ.get("CRISPY_BENCH_DRY_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        print(f"dry run: rank {rank} fails on request", file=sys.stderr)
        sys.exit(3)
    dist, world_reported = _init_dist("gloo", local_rank, world)
    from crispy_amd.sharding import reduce_job_stats, shard_range
    B, T = args.streams, args.frames
    lo, hi = shard_range(world * B, rank, world)
    for _ in range(args.warmup):
        time.sleep(0.005)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.005)
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = _gather_ms(dist, dt * 1e3)
    dt, frames_total = reduce_job_stats(dt, (hi - lo) * T * args.steps)
    if rank == 0:
        print(json.dumps({"metric": "launcher dry run (no GPU work)", "value": None, "unit": None,
                          "n_gpus": world_reported, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "data": "dry-run", "scaling": "weak",
                          "per_rank_ms": per_rank, "frames_total": frames_total,
                          "config": {"workload": "launcher dry run", "streams_per_gpu": B, "frames_per_step": T,
                                     "first_stream_of_rank0": lo}}), flush=True)
    if dist:
        dist.destroy_process_group()


def cfg5(args):
    _numa = _bind_numa(_dist_env()[1], _dist_env()[2])      # before torch / HIP are touched
    line = _cfg5(args, _numa)
    if line is not None:
        print(json.dumps(line), flush=True)


def _cfg5(args, numa, in_process=False):
    """BASELINE configs[4]: Whisper-base full transcribe, 8192 streams sharded across 8 GPUs = 1024 x 30 s clips per
    GPU (static shard by stream id, no data-path collective).  One step = one sub-batch of `--clips` clips resident
    in HBM: log-mel -> encoder -> greedy decode of `--new-tokens` tokens (random-init weights never emit EOT, so the
    decode length is fixed).  value = seconds of audio transcribed per wall second, whole job."""
    import numpy as np
    import torch

    rank, local_rank, world = _dist_env()
    # in_process: a leg of the headline run (cfg2) -- one rank, no process group of its own
    dist, world_reported = (None, 1) if in_process else _init_dist("nccl", local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from crispy_amd.asr import LogMel, WhisperModel
    from crispy_amd.sharding import gather_token_ids, reduce_job_stats, shard_range
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

    hp = HParams.base()
    SUB, NEW = args.clips, args.new_tokens
    depth = max(1, args.cfg5_depth)
    # `depth` sub-batches in flight, each with its own engine handle, HIP stream, workspaces and host thread: the encoder
    # of one sub-batch (matrix cores) runs beside the decode steps of another (cross K|V stream + latency-bound
    # projections).  depth 1 is the serial form whose stage split is printed.
    W = synthetic_whisper_weights(hp, 0)
    lo, hi = shard_range(world * SUB * args.steps, rank, world)      # global clip ids this rank owns
    g = torch.Generator(device=dev).manual_seed(1000 + lo)
    pcm = torch.randn(SUB, 480000, generator=g, device=dev) * 0.1    # every sub-batch reuses one resident buffer
    lens = np.full(SUB, 480000)
    prompt = [50258, 50259, 50359, 50363]
    stage = {"logmel": 0.0, "encoder": 0.0, "decode": 0.0}

    class Lane:
        def __init__(self):
            self.model = WhisperModel(hp, W, device=local_rank)
            self.model.set_precision(args.precision)
            self.lm = LogMel(hp.n_mels, device=local_rank)
            self.melt = torch.zeros(SUB, 3002, hp.n_mels, device=dev)
            self.enc = torch.empty(SUB, 1500, hp.n_audio_state, device=dev)
            self.toks = None

        def step(self, timed):
            t0 = time.perf_counter()
            self.lm.compute_device(pcm.data_ptr(), 480000, lens, 0, self.melt.data_ptr(), stream=0)
            self.lm.synchronize()
            t1 = time.perf_counter()
            self.model.encode_device(self.melt.data_ptr(), SUB, self.enc.data_ptr())
            self.model.synchronize()
            t2 = time.perf_counter()
            self.toks = self.model.decode_greedy_device(self.enc.data_ptr(), SUB, prompt, NEW)
            t3 = time.perf_counter()
            if timed:
                stage["logmel"] += t1 - t0
                stage["encoder"] += t2 - t1
                stage["decode"] += t3 - t2

    lanes = [Lane() for _ in range(depth)]
    torch.cuda.synchronize()             # the fills above ran on torch's stream; the handles use their own

    def barrier():
        if dist:
            dist.barrier()

    for ln in lanes:
        for _ in range(max(1, args.warmup)):
            ln.step(False)
    # the serial stage split: lane 0 alone, nothing beside it
    n_serial = 2
    for _ in range(n_serial):
        lanes[0].step(True)
    stage = {k: v / n_serial for k, v in stage.items()}
    torch.cuda.synchronize()
    barrier()
    if dist:
        _flush_c_stdout()
        barrier()

    def worker(k):
        for _ in range(k, args.steps, depth):
            lanes[k].step(False)

    t0 = time.perf_counter()
    if depth == 1:
        worker(0)
    else:
        import threading
        th = [threading.Thread(target=worker, args=(k,)) for k in range(depth)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    toks = lanes[0].toks
    per_rank = _gather_ms(dist, dt * 1e3, dev)
    dt, clips_total = reduce_job_stats(dt, SUB * args.steps, device=dev)
    # the data product of the job: the fixed-width greedy ids of every rank's last sub-batch, gathered to
    # all ranks in rank order with ONE RCCL all-gather ([SUB, NEW] int32 per rank, ~32 KB: latency-bound) -- outside the
    # timed region, as transcript assembly is in the product; rank 0 prints a checksum of the assembled table
    ids_dev = torch.from_numpy(np.ascontiguousarray(toks[0], dtype=np.int32)).to(dev)
    all_ids = gather_token_ids(ids_dev)
    if rank == 0:
        enc_flops = 87.4e9 * SUB                                  # Whisper-base encoder per 30 s clip
        peak = 157.3 if args.precision == 0 else 2500.0
        ach = enc_flops / stage["encoder"] / 1e12
        line = {
            "metric": "Whisper-base full transcribe RTFx (seconds of audio per wall second, whole job)",
            "value": clips_total * 30.0 / dt, "unit": "x real time (whole job)",
            "n_gpus": world_reported, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 0 else "f16 operands / f32 accumulate", "data": "synthetic",
            "per_rank_ms": per_rank, "process_group": ("nccl (RCCL)" if dist else None), "numa_binding": numa,
            "transcript_ids": {"shape": list(all_ids.shape), "gathered_with": "all_gather_into_tensor" if dist else "single rank",
                               "checksum": int(all_ids.to(torch.int64).sum().item()),
                               "rank0_first_clip": all_ids[0, :8].tolist()},
            "config": {"workload": f"Whisper-base full transcribe (BASELINE configs[4]): {SUB * args.steps} x 30 s clips per "
                                   f"GPU in sub-batches of {SUB}, {NEW} greedy tokens per clip, seeded random-init weights",
                       "clips_per_gpu": SUB * args.steps, "clips_per_step": SUB, "new_tokens": NEW,
                       "sharding": f"clips x{world_reported}, no collective", "clips_per_s": clips_total / dt,
                       "sub_batches_in_flight": depth,
                       "stage_ms_per_step": {k: v * 1e3 for k, v in stage.items()},
                       "serial_step_ms": sum(stage.values()) * 1e3,
                       "tokens_shape": list(np.asarray(toks[0]).shape)},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": None, "kernel": "encoder GEMMs + attention (87.4 GFLOP per clip)",
                         "note": "encoder wall time of rank 0 incl. launch gaps; peak = dense "
                                 + ("f32-input MFMA" if args.precision == 0 else "f16 MFMA")},
        }
    else:
        line = None
    for ln in lanes:
        ln.model.close()
        ln.lm.close()
    if dist:
        dist.destroy_process_group()
    return line


def cfg4(args):
    """BASELINE configs[3]: end-to-end denoise -> adapter scaling / first-frame drop -> s16 WAV hand-off -> 48->16 kHz ->
    30 s chunk -> log-mel -> Whisper-tiny encoder -> greedy decode of `--new-tokens` (64) tokens, `--pipe-streams` (1024)
    streams x 30 s of 48 kHz audio resident in HBM per step, precision mode 1 (the reference engine's arithmetic) unless
    --precision 0.  value = seconds of audio per wall second, whole job.

    RNNoise is a strict 3000-frame recurrence per stream (~42 us per frame and wave whatever the stream count below 4096),
    so inside ONE step it cannot overlap the ASR stages of the same streams; across steps it can: `--pipe-depth 3`
    (default) drives three pipelines (own handles, own HIP streams, own workspaces) from three host threads, and the
    denoise stages of later steps run under the encoder / decoder of earlier ones.  End of round 4, depth 1 / 2 / 3 / 4:
    328 / 321 / 307 / 309 ms per step as a process of its own (8 steps), 324 / 310 / 309 / 323 as a leg of the headline
    process -- which hardware queue a HIP stream lands on is the runtime's choice, and streams that share one take turns
    (NOTEBOOK 8.7).  `--pipe-depth 1` is the serial form whose stage split is printed beside it."""
    rank, local_rank, world = _dist_env()
    numa = _bind_numa(local_rank, world)
    line = _cfg4(args, numa)
    if line is not None:
        print(json.dumps(line), flush=True)


def _cfg4(args, numa, in_process=False):
    rank, local_rank, world = _dist_env()
    import threading

    import numpy as np
    import torch

    dist, world_reported = (None, 1) if in_process else _init_dist("nccl", local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel
    from crispy_amd.pipeline import DenoiseTranscribePipeline
    from crispy_amd.sharding import reduce_job_stats, shard_range
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

    B, T, NEW, depth = args.pipe_streams, 3001, args.new_tokens, max(1, args.pipe_depth)
    weights, weights_label = _rn_weights(args)
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0)
    lo, hi = shard_range(world * B, rank, world)
    x = synth_audio.batch_torch(B, T, dev, first_stream=lo, seed=5).transpose(0, 1).contiguous()      # [B, T, 480]
    prompt = [50258, 50259, 50359, 50363]
    pipes = []
    for _ in range(depth):
        wm = WhisperModel(hp, W, device=local_rank)
        wm.set_precision(args.precision)
        pipes.append(DenoiseTranscribePipeline(weights, wm, B, device=local_rank))
    torch.cuda.synchronize()
    for p in pipes:                       # warm-up: workspaces, captured decode graphs, f16 weight copies
        for _ in range(max(1, args.warmup)):
            toks, _ = p.run(x, prompt, NEW)
            p.ds.reset()
    toks, _ = pipes[0].run(x, prompt, NEW)   # one more run of pipeline 0 alone, everything warm: the serial stage split
    pipes[0].ds.reset()
    stage = dict(pipes[0].timings)
    serial_ms = sum(stage.values()) * 1e3

    def barrier():
        if dist:
            dist.barrier()

    def worker(k):
        for i in range(k, args.steps, depth):
            pipes[k].run(x, prompt, NEW)
            pipes[k].ds.reset()            # every step is a fresh batch of streams

    torch.cuda.synchronize()
    barrier()
    if dist:
        _flush_c_stdout()
        barrier()
    t0 = time.perf_counter()
    if depth == 1:
        worker(0)
    else:
        th = [threading.Thread(target=worker, args=(k,)) for k in range(depth)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    per_rank = _gather_ms(dist, dt * 1e3, dev)
    dt, streams_total = reduce_job_stats(dt, B * args.steps, device=dev)
    if rank == 0:
        audio_s = streams_total * 30.0
        line = {
            "metric": "end-to-end denoise -> Whisper-tiny greedy decode RTFx (seconds of 48 kHz audio per wall second, whole job)",
            "value": audio_s / dt, "unit": "x real time (whole job)", "n_gpus": world_reported, "steps": args.steps,
            "warmup": max(1, args.warmup), "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (RNNoise, resampler, log-mel) + " + ("f32" if args.precision == 0 else "f16 operands / f32 accumulate") + " (Whisper)",
            "data": "synthetic", "per_rank_ms": per_rank, "process_group": ("nccl (RCCL)" if dist else None), "numa_binding": numa,
            "config": {"workload": f"BASELINE configs[3]: {B} streams x 30 s @ 48 kHz per GPU and step -> RNNoise -> WAV s16 -> 48->16 kHz "
                                   f"-> log-mel -> Whisper-tiny encoder -> {NEW} greedy tokens; {weights_label}; seeded random-init Whisper-tiny",
                       "streams_per_gpu": B, "new_tokens": NEW, "precision_mode": args.precision, "pipe_depth": depth,
                       "serial_step_ms": serial_ms, "serial_stage_ms": {k: v * 1e3 for k, v in stage.items()},
                       "sharding": f"streams x{world_reported}, no collective",
                       "tokens_checksum": int(np.asarray(toks, dtype=np.int64)[toks >= 0].sum())},
            "roofline": {"bound": "mfma", "achieved": encoder_flops(hp) * streams_total / dt / 1e12,
                         "peak": 157.3 if args.precision == 0 else 2500.0, "unit": "TFLOP/s",
                         "frac": encoder_flops(hp) * streams_total / dt / 1e12 / (157.3 if args.precision == 0 else 2500.0),
                         "traffic": None, "kernel": "Whisper encoder GEMMs + attention, priced against the WHOLE step "
                                                    "(RNNoise, resampler, log-mel and the decoder included in the time)"},
        }
    else:
        line = None
    for p in pipes:
        p.close()
        p.whisper.close()
    if dist:
        dist.destroy_process_group()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("cfg2", "cfg4", "cfg5"), default="cfg2",
                    help="cfg2 = batched RNNoise (BASELINE configs[1], the headline); cfg4 = end-to-end denoise -> WAV -> "
                         "48->16 kHz -> Whisper-tiny greedy decode (BASELINE configs[3]); cfg5 = Whisper-base full "
                         "transcribe shards (BASELINE configs[4])")
    ap.add_argument("--host-fed", action="store_true",
                    help="cfg2: feed every step from page-locked HOST memory through crispy_rn_process (copy-in, kernels and "
                         "copy-out pipelined per rank) -- the PCIe-inclusive rate, labelled as such; never the headline")
    ap.add_argument("--rnnoise-model", default=None,
                    help="cfg2 / cfg4: an rnnoise-nu text model file (what nnnoiseless::RnnModel::from_read parses) instead "
                         "of the seeded synthetic int8 weights")
    ap.add_argument("--ggml", default=None,
                    help="ASR leg of cfg2: a whisper.cpp GGML model file (e.g. ggml-tiny.bin) instead of the seeded "
                         "random-init Whisper-tiny (SURVEY.md 8d cfg 3)")
    ap.add_argument("--streams", type=int, default=4096, help="cfg2: streams per GPU")
    ap.add_argument("--frames", type=int, default=100, help="cfg2: frames per stream per step")
    ap.add_argument("--clips", type=int, default=256, help="cfg5: 30 s clips per step (sub-batch) per GPU")
    ap.add_argument("--new-tokens", type=int, default=None, help="greedy tokens per clip (default: cfg5 32, cfg4 64)")
    ap.add_argument("--pipe-streams", type=int, default=1024, help="cfg4: 48 kHz streams per GPU (30 s each)")
    ap.add_argument("--cfg5-depth", type=int, default=2,
                    help="cfg5: sub-batches in flight (own engine handle, HIP stream and host thread each): the encoder of one "
                         "runs beside the decode steps of another; 1 = serial")
    ap.add_argument("--pipe-depth", type=int, default=3,
                    help="cfg4: pipelines in flight (own handles and HIP streams, one host thread each): with 2, RNNoise of "
                         "step k + 1 runs under the encoder / decoder of step k; 1 = serial")
    ap.add_argument("--precision", type=int, default=1,
                    help="cfg5: 1 = f16 operands / f32 accumulation, the arithmetic of the reference's whisper.cpp engine (default); "
                         "0 = exact f32 products (the mode the oracle parity is stated in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="cfg2: skip the PCIe-inclusive leg (f32 and int16 transport) of the default run")
    ap.add_argument("--no-asr", action="store_true", help="skip the Whisper-tiny leg of the metric")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-stream process_frame latency leg")
    ap.add_argument("--no-cfg45", action="store_true", help="cfg2: skip the in-process cfg4 / cfg5 legs")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="cfg2: length of the sustained-rate loop behind the timed steps (0 = off)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes; roofline.traffic then comes from profiles/")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test on CPU (gloo, no GPU work, no value); used by tests/test_sharding_gloo.py")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"cfg5": 4, "cfg4": 2}.get(args.workload, 10)
    if args.new_tokens is None:
        args.new_tokens = 64 if args.workload == "cfg4" else 32
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # One process per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process IS a rank; without
    # one, --gpus N > 1 starts its own N ranks as fresh children BEFORE anything here has imported torch or touched
    # the GPU (no exec), relays rank 0's JSON line and exits with the job's code.  N = 1 stays in-process, so
    # `rocprofv3 ... -- python3 bench.py` profiles a single process.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            from crispy_amd.launch import spawn_ranks
            sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
              "refusing to report one as the other", file=sys.stderr)
        sys.exit(2)
    if args.pmc_child:
        return pmc_child(args)
    if args.dry_run:
        return dry_run(args)
    if args.workload == "cfg5":
        return cfg5(args)
    if args.workload == "cfg4":
        return cfg4(args)
    return cfg2(args)


def _rn_weights(args):
    """Seeded synthetic int8 weights, or the path of an rnnoise-nu model file (parsed by crispy_rn_create_from_file)."""
    if args.rnnoise_model:
        return args.rnnoise_model, f"rnnoise-nu model file {os.path.basename(args.rnnoise_model)}"
    from crispy_amd import synthetic_weights
    return synthetic_weights(0), "seeded synthetic int8 weights"


def _bind_numa(local_rank: int, world: int):
    """Before torch or HIP are touched: pin this rank to the CPUs of its GPU's NUMA node (crispy_amd/launch.py)."""
    from crispy_amd.launch import bind_rank_to_gpu_numa
    return bind_rank_to_gpu_numa(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))


def host_fed_leg(ds, d_in, B, T, steps=20, warmup=2):
    """The PCIe-inclusive flavour of the headline, inside the default run: the same step fed from
    page-locked HOST memory through crispy_rn_process (f32 samples) and crispy_rn_process_s16 (int16 samples: the formats the
    reference's capture / recording paths hold, audio.rs:794-855) -- copy-in, kernels and copy-out pipelined in pieces of
    frames.  717 k "streams" of the HBM-resident headline are 275 GB/s of f32 PCM each way: no host link feeds that; this is
    what one does."""
    import numpy as np
    from crispy_amd.denoise import DenoiseState
    out = {}
    h_f = d_in.cpu().numpy()
    legs = (("f32", h_f, lambda x, o, v: ds.process_into(x, o, v)),
            ("s16", np.clip(np.rint(h_f), -32768, 32767).astype(np.int16), lambda x, o, v: ds.process_s16(x, out=o, vad=v)))
    for name, h_in, call in legs:
        h_out = np.empty_like(h_in)
        h_vad = np.empty((T, B), dtype=np.float32)
        for arr in (h_in, h_out):
            DenoiseState.register_host(arr)
        try:
            for _ in range(warmup):
                call(h_in, h_out, h_vad)
            t0 = time.perf_counter()
            for _ in range(steps):
                call(h_in, h_out, h_vad)          # (returns when `out` is complete: nothing left in flight)
            dt = time.perf_counter() - t0
        finally:
            for arr in (h_in, h_out):
                DenoiseState.unregister_host(arr)
        out[name] = {"value": B * T * steps / dt / 100.0, "ms_per_step": dt / steps * 1e3, "steps": steps,
                     "pcie_gbps_each_way": h_in.nbytes * steps / dt / 1e9, "bytes_per_sample": int(h_in.itemsize)}
    return {"unit": "concurrent real-time 48 kHz streams (this GPU), input and output in page-locked host memory",
            "value": out["f32"]["value"], "ms_per_step": out["f32"]["ms_per_step"],
            "pcie_gbps_each_way": out["f32"]["pcie_gbps_each_way"], "f32": out["f32"], "s16": out["s16"],
            "s16_over_f32": out["s16"]["value"] / out["f32"]["value"],
            "entry_points": "crispy_rn_process / crispy_rn_process_s16 (pieces of frames on three streams; DESIGN section 6)"}


def cfg2(args):
    rank, local_rank, world = _dist_env()
    numa = _bind_numa(local_rank, world)
    import torch

    dist, world_reported = _init_dist("nccl", local_rank, world)
    use_dist = dist is not None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from crispy_amd import synthetic_weights, synth_audio
    from crispy_amd.denoise import DenoiseState

    B, T = args.streams, args.frames
    weights, weights_label = _rn_weights(args)
    ds = DenoiseState(weights, B, local_rank)   # fails loudly without libcrispy_hip.so / gfx950
    # streams shard by stream id (crispy_amd.sharding: block partition, no data-path collective):
    # weak scaling, rank r owns global stream ids [r*B, (r+1)*B)
    from crispy_amd.sharding import reduce_job_stats, shard_range
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    d_in = synth_audio.batch_torch(B, T, dev, first_stream=lo, seed=0)
    d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()
    h_in = h_out = h_vad = None
    if args.host_fed:
        # the input lives in page-locked HOST memory of this rank (first-touched after the NUMA binding above) and every
        # step crosses PCIe both ways: crispy_rn_process cuts the call into pieces and overlaps copy-in / kernels / copy-out
        import numpy as np
        h_in = d_in.cpu().numpy()
        h_out = np.empty_like(h_in)
        h_vad = np.empty((T, B), dtype=np.float32)
        for arr in (h_in, h_out):
            DenoiseState.register_host(arr)

    def step():
        if args.host_fed:
            ds.process_into(h_in, h_out, h_vad)
        else:
            ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T)

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ds.synchronize()
    torch.cuda.synchronize()
    barrier()
    if use_dist:
        _flush_c_stdout()
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ds.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = _gather_ms(dist, dt * 1e3, dev)
    # max time over ranks, frames summed over ranks (two tiny RCCL all-reduces)
    dt, frames_total = reduce_job_stats(dt, B * T * args.steps, device=dev)

    # dominant-kernel duration, measured live with hipEvents on the launch stream
    ds.set_timing(True)
    k_ms = []
    for _ in range(3):
        step()
        ds.synchronize()
        k_ms.append(ds.last_kernel_ms())
    ds.set_timing(False)
    # A sustained figure beside the driver-timed burst (20 steps = 0.12 s): the same step for `--sustain-seconds` of wall
    # time (default 5), clock sampled by a side thread.  Synchronised every 50 steps so that the host cannot run ahead.
    sustained = None
    if args.sustain_seconds > 0 and not args.host_fed:
        with _ClockSampler(local_rank) as clk:
            n_sus = 0
            t0s = time.perf_counter()
            while time.perf_counter() - t0s < args.sustain_seconds:
                for _ in range(50):
                    step()
                ds.synchronize()
                n_sus += 50
            dts = time.perf_counter() - t0s
        sustained = {"seconds": dts, "steps": n_sus, "ms_per_step": dts / n_sus * 1e3,
                     "value": B * T * n_sus / dts / 100.0, "unit": "concurrent real-time 48 kHz streams (this rank)",
                     "sclk_mhz_mean": clk.mean(), "sclk_samples": len(clk.samples),
                     "sclk_source": (clk.path + " (matched by PCI address)") if clk.by_pci else
                                    ("highest of " + str(len(clk.paths)) + " pp_dpm_sclk files" if clk.paths else "no pp_dpm_sclk in sysfs")}
    from crispy_amd import _native as N
    # rn_frame_kernel launches per step (a call starts with short launches of 3 and 8 frames, then 12 per launch);
    # per-launch figures below are averages over them: algorithmic bytes of a step / launches, kernel time / launches
    launches = N.lib().crispy_rn_n_launches(T)
    frame_ms = sum(k[0] for k in k_ms) / len(k_ms) / launches
    total_ms = sum(k[1] for k in k_ms) / len(k_ms)
    if args.host_fed:
        d_out.copy_(torch.from_numpy(h_out))
    finite = bool(torch.isfinite(d_out).all().item())

    fps = frames_total / dt
    if rank == 0:
        alg_bytes = BYTES_PER_STREAM_FRAME * B * T / launches
        # HBM traffic per launch: measured live on this box by two rocprofv3 --pmc child passes (measure_traffic);
        # if that is not possible the committed passes of the same launch shape are quoted and labelled as such.
        traffic = None
        traffic_src = None
        valu = None
        if world == 1 and not args.no_live_traffic:
            try:
                tr = measure_traffic(B, T)
                traffic = int((tr["fetch_bytes"] + tr["write_bytes"]) * B * T / launches)
                traffic_src = {"how": "live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                                      "`bench.py --pmc-child` on this box; FETCH_SIZE x2 (gfx950), KiB -> bytes",
                               "fetch_bytes_per_stream_frame": tr["fetch_bytes"],
                               "write_bytes_per_stream_frame": tr["write_bytes"]}
            except Exception as e:
                traffic_src = {"how": "committed profile (live measurement failed)", "error": str(e)[:200]}
        try:
            pm_file, streams, psf = committed_pmc()
            if streams == B:
                sf = B * T / launches                        # stream-frames per (average) launch
                if traffic is None:
                    traffic = int((2 * psf["FETCH_SIZE"] + psf["WRITE_SIZE"]) * 1024 * sf)
                    traffic_src = dict(traffic_src or {}, file=pm_file)
                # what actually bounds this kernel: VALU issue slots.  SQ_ACTIVE_INST_VALU counts 4-cycle issue
                # slots; 1024 SIMDs; priced against the live kernel time at the 2.4 GHz peak clock.
                valu = {"insts_per_stream_frame": round(psf["SQ_INSTS_VALU"]),
                        "issue_frac": psf["SQ_ACTIVE_INST_VALU"] * sf * 4 / (1024 * frame_ms * 1e-3 * 2.4e9),
                        "source": f"{pm_file} (rocprofv3 --pmc SQ_ACTIVE_INST_VALU), live kernel time"}
        except Exception:
            pass
        achieved = alg_bytes / (frame_ms * 1e-3) / 1e9
        flops = FLOPS_PER_STREAM_FRAME * B * T / launches / (frame_ms * 1e-3) / 1e12
        line = {
            "metric": "concurrent real-time 48 kHz streams/GPU (RNNoise)"
                      + (" -- HOST-FED flavour: PCIe copies inside the timed region, not the headline" if args.host_fed else ""),
            "value": fps / 100.0,
            "unit": "concurrent real-time 48 kHz streams (whole job)",
            "n_gpus": world_reported, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (", host-resident (PCIe-inclusive)" if args.host_fed else ""),
            "per_rank_ms": per_rank_ms,
            "process_group": ("nccl (RCCL)" if use_dist else None), "numa_binding": numa,
            "config": {"workload": f"Batched RNNoise: {B} concurrent 48 kHz mono streams per GPU x {T} frames per step "
                                   f"(BASELINE configs[1]), {weights_label}"
                                   + (", fed from page-locked host memory through crispy_rn_process" if args.host_fed else ""),
                       "streams_per_gpu": B, "frames_per_step": T, "sharding": f"streams x{world_reported}, no collective",
                       "frames_per_s": fps, "output_finite": finite},
            # with the per-stream state on chip this kernel is bound by the fp32 vector pipe (117 flop per
            # algorithmic byte against a machine balance of 20), so that is the roof `frac` is taken against; the HBM
            # figures (algorithmic bytes per launch against 8 TB/s, counter traffic) ride beside it.
            "roofline": {"bound": "valu", "achieved": flops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / VALU_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rn_frame_kernel", "kernel_ms": frame_ms, "launches_per_step": launches,
                         "enqueue_ms": total_ms,
                         "alg_flops_per_launch": FLOPS_PER_STREAM_FRAME * B * T / launches,
                         "alg_bytes_per_launch": alg_bytes, "valu": valu,
                         "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                 "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None}},
            "sustained": sustained,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.host_fed and not args.no_host_fed:
            try:
                line["host_fed"] = host_fed_leg(ds, d_in, B, T)
            except Exception as e:     # a reported extra, never a reason to lose the headline
                line["host_fed"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_latency:
            # the literal drop-in: ONE stream, ONE frame per call, host slices (audio.rs:260-268), from a C program
            # compiled against include/crispy_hip.h; a child process, timed call by call
            try:
                from tests import c_dropin
                x1 = (synth_audio.stream_np(11, 20, silent=False) * 32768.0).astype("float32").reshape(20, 480)
                lat_w = weights if not isinstance(weights, str) else synthetic_weights(0)
                line["latency_us"] = c_dropin.run(lat_w, x1, timed_calls=10000)[2]
            except Exception as e:     # a reported extra, never a reason to lose the headline
                line["latency_us"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_asr:
            del d_in, d_out
            torch.cuda.empty_cache()
            line["asr"] = asr_leg(local_rank, ggml_path=args.ggml)
            # the second half of BASELINE.json's metric at the top level, in the precision that ships (mode 1 = the
            # reference engine's arithmetic; the Rust binding's default), with the mode-0 figure beside it
            a = line["asr"]
            line["asr_rtfx"] = {"metric": "Whisper-tiny RTFx at 1/8 GPU (one MI355X of the node)",
                                "value": a["f16_operand_mode"]["rtfx_end_to_end"], "unit": "x real time",
                                "precision_mode": 1, "arithmetic": "f16 operands, f32 accumulation (whisper.cpp / ggml mul_mat)",
                                "clips": a["clips"], "new_tokens": a["new_tokens"],
                                "value_precision_mode_0": a["rtfx_end_to_end"],
                                "single_30s_chunk_from_host_ms": a["single_clip"]["f16_operand_mode_ms"],
                                "model": a["model"]}
        if world == 1 and not args.no_cfg45:
            # BASELINE configs[3] and configs[4] inside the same run (and so inside the driver's clock around it): one
            # 1024-stream end-to-end step pair and 256-clip Whisper-base steps, each through its own workload function
            import copy
            ds.close()
            torch.cuda.empty_cache()
            for key, fn, over in (("cfg4", _cfg4, dict(steps=6, warmup=1, new_tokens=64)),
                                  ("cfg5", _cfg5, dict(steps=2, warmup=1, new_tokens=32))):
                a2 = copy.copy(args)
                for k2, v2 in over.items():
                    setattr(a2, k2, v2)
                try:
                    t0c = time.perf_counter()
                    r = fn(a2, numa, in_process=True)
                    c = r["config"]
                    line[key] = {"workload": c["workload"], "ms_per_step": r["ms_per_step"], "rtfx": r["value"], "steps": r["steps"],
                                 "stage_ms": c.get("serial_stage_ms") or c.get("stage_ms_per_step"),
                                 "serial_step_ms": c.get("serial_step_ms"), "pipe_depth": c.get("pipe_depth"),
                                 "precision_mode": a2.precision, "tokens_checksum": c.get("tokens_checksum")
                                 or (r.get("transcript_ids") or {}).get("checksum"), "leg_wall_s": time.perf_counter() - t0c}
                    st = line[key]["stage_ms"] or {}
                    if key == "cfg4" and st.get("resample"):
                        # the 48 -> 16 kHz stage against its two roofs: the bytes it must move (f32 in at 48 kHz, f32 out at 16 kHz)
                        # and the f16 matrix-core flops of the pair form it runs as (3 products of 1404 x 342 x 2112 per 30 s stream)
                        n_str = c.get("streams_per_gpu") or 1024
                        by = n_str * (1440000 + 480168) * 4.0
                        fl = n_str * 3 * 2.0 * 1404 * 342 * 2112
                        ms = st["resample"]
                        line[key]["resample_roofline"] = {
                            "hbm": {"achieved": by / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": by / ms / 1e6 / 8000.0},
                            "mfma": {"achieved": fl / ms / 1e9, "peak": 2500.0, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / 2500.0},
                            "note": "rubato's FftFixedIn as a fixed linear map on the f16 matrix cores with f16 (hi, lo) operand pairs (DESIGN section 4)"}
                except Exception as e:     # a reported extra, never a reason to lose the headline
                    line[key] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
