"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path behind the C ABI against the
CPU oracle on identical seeded inputs, against the committed golden vectors, and through
size-independent properties at BASELINE size.

Tolerance (BASELINE.json north_star: denoised PCM within 1e-4 relative): per stream,
max |gpu - oracle| <= 1e-4 * max |oracle|  (+ 1e-3 absolute, samples are in int16 range)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "3"], ids=["one_wave_per_stream", "three_wave_stage_pipeline"])
def rn_waves(request, monkeypatch):
    """Every test of this file runs against both forms of the frame kernel: one wave per stream (what a handle picks above
    1280 streams) and the three-wave stage pipeline (what it picks up to there); CRISPY_RN_WAVES is read at create time."""
    monkeypatch.setenv("CRISPY_RN_WAVES", request.param)
    return request.param

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rnnoise_golden.npz")
REL = 1e-4


def _mk(weights, B):
    from crispy_amd.denoise import DenoiseState
    return DenoiseState(weights, B, 0)   # raises if libcrispy_hip.so or the gfx950 device is missing


def _assert_pcm_close(out, ref, what=""):
    peak = np.abs(ref).max()
    err = np.abs(out - ref).max()
    assert err <= REL * peak + 1e-3, f"{what}: err {err} vs peak {peak}"


PITCH_TOL = 1e-5      # a decision whose gap exceeds this (units: oracle/rnnoise_oracle.c, margin_note) must come out the same


def _account_for_pitch(got, ref, margin, what):
    """Pitch index, an integer: EQUAL to the oracle's on every frame whose decision margin exceeds PITCH_TOL -- the smallest gap
    at any comparison that decides the index (top-two ranking of find_best_pitch, the interpolation tests, remove_doubling's
    g1 - thresh), in units in which two correct f32 implementations differ by ~1e-6 -- or whose PREVIOUS frame already
    differed (remove_doubling's continuity bonus reads last_period: a frame inherits its predecessor's near tie).  Returns
    (mismatches, frames with a margin inside the tolerance); the callers bound the rate."""
    got, ref = np.asarray(got), np.asarray(ref)
    diff = got != ref
    inherited = np.zeros_like(diff)
    inherited[1:] = diff[:-1]
    unexplained = diff & (margin > PITCH_TOL) & ~inherited
    assert not unexplained.any(), (what, "frames", np.nonzero(unexplained)[0].tolist(), "kernel", got[unexplained].tolist(),
                                   "oracle", ref[unexplained].tolist(), "margins", margin[unexplained].tolist())
    return int(diff.sum()), int((margin <= PITCH_TOL).sum())


def test_native_library_is_the_one_running():
    from crispy_amd import _native as N
    assert os.path.exists(N.LIB_PATH)
    assert N.lib().crispy_device_count() >= 1
    import subprocess
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libcrispy_hip.so" in maps


@pytest.mark.parametrize("case", ["seed0", "seed1", "seed2", "silence", "tone440", "whisper_quiet"])
def test_golden_vectors(case):
    G = np.load(GOLD)
    w = G["weights" + case[-1]] if case.startswith("seed") else G["weights0"]
    x = G[f"{case}/x"]
    ds = _mk(w, 1)
    out, vad = ds.process(x[:, None, :])
    _assert_pcm_close(out[:, 0], G[f"{case}/out"], case)
    assert np.abs(vad[:, 0] - G[f"{case}/vad"]).max() < 1e-4


def test_level_extremes_through_the_fixed_point_gain_network(oracle, weights0):
    """The gain network runs on int8 MFMAs with the activations as per-vector fixed point (rn_kernels.hip,
    RN_GRU_MFMA == 2): streams at full scale (clipping square wave, +-32768 noise), with a large DC offset, at
    1e-3 of full scale and a loud burst after near-silence must all stay inside the PCM tolerance."""
    rng = np.random.default_rng(7)
    T = 40
    n = T * 480
    t = np.arange(n) / 48000.0
    sig = [
        32767.0 * np.sign(np.sin(2 * np.pi * 173.0 * t)),                       # clipping square wave
        rng.uniform(-32768, 32767, n),                                          # full-scale noise
        20000.0 + 3000.0 * np.sin(2 * np.pi * 220.0 * t) + 500 * rng.standard_normal(n),   # DC offset
        30.0 * np.sin(2 * np.pi * 300.0 * t) + 5.0 * rng.standard_normal(n),    # 1e-3 of full scale
        np.where(t < 0.2, 0.5 * rng.standard_normal(n), 25000.0 * np.sin(2 * np.pi * 140.0 * t)),  # burst
    ]
    x = np.stack(sig, axis=0).astype(np.float32).reshape(len(sig), T, 480).transpose(1, 0, 2).copy()
    ds = _mk(weights0, len(sig))
    out, vad = ds.process(x)
    assert np.isfinite(out).all()
    for b in range(len(sig)):
        ref, rvad = oracle.OracleDenoiseState(weights0).process(x[:, b])
        _assert_pcm_close(out[:, b], ref, f"level case {b}")
        assert np.abs(vad[:, b] - rvad).max() < 1e-4, b


def test_parity_mixed_batch_with_taps(oracle, weights0):
    """12 different streams x 60 frames incl. a silent stream and the cfg-1 clip: PCM, VAD, features,
    gains and pitch against the oracle."""
    import torch
    from crispy_amd import synth_audio as SA
    B, T = 12, 60
    x = SA.batch_np(B, T) * np.float32(32768.0)
    x[:, 1] = SA.cfg1_clip(T).reshape(T, 480) * 32768.0
    ds = _mk(weights0, B)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    d_taps = torch.zeros(T, B, 72, device=dev)
    d_vad = torch.zeros(T, B, device=dev)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad.data_ptr(), d_taps.data_ptr())
    ds.synchronize()
    out, taps, vad = d_out.cpu().numpy(), d_taps.cpu().numpy(), d_vad.cpu().numpy()
    n_diff = n_close = 0
    per_stream = []
    for b in range(B):
        ro, rv, rt, margin = oracle.OracleDenoiseState(weights0).process(x[:, b], with_taps=True, with_margin=True)
        _assert_pcm_close(out[:, b], ro, f"stream {b}")
        assert np.abs(vad[:, b] - rv).max() < 1e-4
        assert np.abs(taps[:, b, 42:64] - rt[:, 42:64]).max() < 1e-4          # gains
        assert np.abs(taps[:, b, :42] - rt[:, :42]).max() <= 1e-4 * max(1.0, np.abs(rt[:, :42]).max())
        assert np.array_equal(taps[:, b, 67], rt[:, 67])                      # silence flags
        d, c = _account_for_pitch(taps[:, b, 64], rt[:, 64], margin, f"stream {b}")
        per_stream.append(d)
        n_diff += d
        n_close += c
    print(f"pitch index: {n_diff} of {B * T} frames differ from the oracle (per stream {per_stream}), every one inside the decision "
          f"margin {PITCH_TOL:g} or behind a frame that was; {n_close} frames have a margin that small")
    assert n_diff <= 0.002 * B * T + 1, (n_diff, per_stream)


def test_highpass_stage_is_bit_exact(oracle, weights0):
    """The high-pass biquad (f32 state rounding, f64 products: Appendix A.3 step 1) is the one stage of the frame that
    is reproduced bit for bit -- the kernel keeps the reference's rounding points (products exact in f64, one rounding
    per sub/add, f32 state), only fused and prefetched differently.  Checked on the last frame's high-passed signal of
    five streams after 33 frames (debug capture slot 3824..4304)."""
    from crispy_amd import synth_audio as SA
    B, T = 5, 33
    x = SA.batch_np(B, T) * np.float32(32768.0)
    ds = _mk(weights0, B)
    ds.debug_capture(True)
    ds.process(x)
    for b in range(B):
        st = oracle.OracleDenoiseState(weights0)
        st.process(x[:, b])
        ref = st.debug()[3824:4304]
        got = ds.debug_read(b)[3824:4304]
        assert np.abs(ref).max() > 100.0
        assert np.array_equal(got, ref), f"stream {b}: high-passed frame differs from the oracle"


def test_cfg1_single_clip_through_the_adapter(oracle, weights0):
    """BASELINE cfg 1 (shortened to 3 s): the RnnNoiseProcessor adapter (x32768, clamp, volume,
    first-frame drop: audio.rs:261-278) over the HIP path equals the same adapter over the oracle."""
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import RnnNoiseProcessor
    T = 300
    clip = SA.cfg1_clip(T)
    # batched host call = what push_sample does frame by frame
    ds = _mk(weights0, 1)
    out, _ = ds.process((clip * np.float32(32768.0)).reshape(T, 1, 480))
    got = (np.clip(out[:, 0] / np.float32(32768.0), -1, 1) * np.float32(0.8))[1:]
    ro, _ = oracle.OracleDenoiseState(weights0).process(clip * np.float32(32768.0))
    want = (np.clip(ro / np.float32(32768.0), -1, 1) * np.float32(0.8))[1:]
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-7
    # and literally sample by sample for the first 5 frames
    proc = RnnNoiseProcessor(weights0, 48000.0, 48000.0, 0.8, 1, 0)
    chunks = []
    for s in clip[:480 * 5]:
        r = proc.push_sample([s])
        if r is not None:
            chunks.append(r[:, 0])
    got2 = np.concatenate(chunks)
    assert got2.shape == (480 * 4,)     # first frame dropped
    assert np.abs(got2 - want[:4].ravel()).max() <= 1e-4 * np.abs(want).max() + 1e-7


def test_chunked_calls_equal_one_call(weights0):
    """State carried across calls (history roll, GRU/cepstral state): ragged call sizes,
    including ones shorter than the 4-frame history and ones that cross the internal 250-frame chunk."""
    from crispy_amd import synth_audio as SA
    B, T = 5, 300
    x = SA.batch_np(B, T) * np.float32(32768.0)
    a, va = _mk(weights0, B).process(x)
    ds = _mk(weights0, B)
    outs, vads, t0 = [], [], 0
    for n in (1, 2, 3, 1, 7, 250, 36):
        o, v = ds.process(np.ascontiguousarray(x[t0:t0 + n]))
        outs.append(o); vads.append(v); t0 += n
    assert t0 == T
    assert np.array_equal(np.concatenate(outs), a) and np.array_equal(np.concatenate(vads), va)


def test_layouts_and_host_device_entry_points_agree(weights0):
    import torch
    from crispy_amd import synth_audio as SA
    B, T = 6, 9
    x = SA.batch_np(B, T) * np.float32(32768.0)
    a, va = _mk(weights0, B).process(x, "tbf")
    b, vb = _mk(weights0, B).process(np.ascontiguousarray(x.transpose(1, 0, 2)), "btf")
    assert np.array_equal(a, b.transpose(1, 0, 2)) and np.array_equal(va, vb)
    ds = _mk(weights0, B)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T)
    ds.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), a)


def test_pipelined_host_path_equals_device_path(weights0):
    """Calls above 8 MB take the pipelined host path (pieces of frames on three streams, copy-out on its own thread):
    both layouts, a piece count that does not divide the call, pageable and registered buffers, VAD included, must
    reproduce the device entry point bit for bit -- twice in a row (events and staging are reused)."""
    import torch
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import DenoiseState
    B, T = 256, 37                        # 18 MB per direction: the pipelined path with a single piece
    x = SA.batch_np(B, T) * np.float32(32768.0)
    ref_ds = _mk(weights0, B)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty(T, B, device="cuda")
    torch.cuda.synchronize()
    ref_ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad=d_vad.data_ptr())
    ref_ds.synchronize()
    ref, rvad = d_out.cpu().numpy(), d_vad.cpu().numpy()
    a, va = _mk(weights0, B).process(x, "tbf")
    assert np.array_equal(a, ref) and np.array_equal(va, rvad)
    # many pieces: 4096 streams x 30 frames = 236 MB per direction -> pieces of 11 frames (11, 11, 8)
    B2, T2 = 4096, 30
    x2 = SA.batch_np(B2, T2) * np.float32(32768.0)
    ds_dev = _mk(weights0, B2)
    d_in = torch.from_numpy(x2).cuda()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty(T2, B2, device="cuda")
    torch.cuda.synchronize()
    ds_dev.process_device(d_in.data_ptr(), d_out.data_ptr(), T2, d_vad=d_vad.data_ptr())
    ds_dev.synchronize()
    ref2, rvad2 = d_out.cpu().numpy(), d_vad.cpu().numpy()
    del d_in, d_out, ds_dev
    ds = _mk(weights0, B2)
    out = np.empty_like(x2)
    vad = np.empty((T2, B2), np.float32)
    ds.process_into(x2, out, vad, "tbf")
    assert np.array_equal(out, ref2) and np.array_equal(vad, rvad2)
    # the next call continues the streams: compare with a fresh handle fed both halves through the other layout
    xb = np.ascontiguousarray(x2.transpose(1, 0, 2))
    ob = np.empty_like(xb)
    ds_b = _mk(weights0, B2)
    ds_b.process_into(xb, ob, vad, "btf")
    assert np.array_equal(ob.transpose(1, 0, 2), ref2) and np.array_equal(vad, rvad2)
    for arr in (x2, out, vad):
        DenoiseState.register_host(arr)
    try:
        ds_r = _mk(weights0, B2)
        out[:] = 0
        ds_r.process_into(x2, out, vad, "tbf")
        assert np.array_equal(out, ref2) and np.array_equal(vad, rvad2)
    finally:
        for arr in (x2, out, vad):
            DenoiseState.unregister_host(arr)


def test_int16_transport_equals_the_f32_entry_point_then_the_wav_quantisation(weights0):
    """`crispy_rn_process_s16*`: int16 PCM in and out -- the formats the reference's capture and
    recording paths hold (audio.rs:794-855, commands/transcription.rs:306-313).  Integer work, so the bar is equality:
    s16 entry point == f32 entry point on float(s), then the oracle's `wav_s16_roundtrip` (/ 32768, clamp, x 32767
    truncated: audio.rs:270-273, recording.rs:109-110) -- at 4096 streams, both layouts, host (small call, pipelined
    call) and device entry points, chunked calls continuing the streams, VAD identical."""
    import torch
    from crispy_amd import synth_audio as SA
    from oracle import resample_oracle as RO
    B, T = 4096, 24
    xf = SA.batch_np(B, T) * np.float32(32768.0)
    xf[:, 7] *= 3.0                                                   # a stream that clips at the int16 rails
    xi = np.clip(np.rint(xf), -32768, 32767).astype(np.int16)         # [T, B, 480]
    ref_ds = _mk(weights0, B)
    ref, rvad = ref_ds.process(xi.astype(np.float32))
    want = (RO.wav_s16_roundtrip(ref / np.float32(32768.0)) * np.float32(32768.0)).astype(np.int16)
    assert np.abs(want).max() > 20000 and len(np.unique(want)) > 10000
    # host entry point: 94 MB in -> the pipelined path; then two chunked calls on a fresh handle
    out, vad = _mk(weights0, B).process_s16(xi)
    assert out.dtype == np.int16 and np.array_equal(out, want) and np.array_equal(vad, rvad)
    ds = _mk(weights0, B)
    o1, v1 = ds.process_s16(xi[:10])
    o2, v2 = ds.process_s16(xi[10:])
    assert np.array_equal(np.concatenate([o1, o2]), want) and np.array_equal(np.concatenate([v1, v2]), rvad)
    # stream-major layout through the device entry point
    ds = _mk(weights0, B)
    d_in = torch.from_numpy(np.ascontiguousarray(xi.transpose(1, 0, 2))).cuda()
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty(T, B, device="cuda")
    torch.cuda.synchronize()
    ds.process_s16_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad=d_vad.data_ptr(), layout="btf")
    ds.synchronize()
    assert np.array_equal(d_out.cpu().numpy().transpose(1, 0, 2), want) and np.array_equal(d_vad.cpu().numpy(), rvad)
    # a handle may mix the transports: f32 call, then int16 call, continue one another's streams
    ds = _mk(weights0, B)
    a, _ = ds.process(xi[:10].astype(np.float32))
    b, _ = ds.process_s16(xi[10:])
    assert np.array_equal(a, ref[:10]) and np.array_equal(b, want[10:])
    # the single-stream drop-in shape (a small call: one copy in, one out)
    one = _mk(weights0, 1)
    o, _ = one.process_s16(np.ascontiguousarray(xi[:, 5:6]))
    assert np.array_equal(o[:, 0], want[:, 5])


def test_streams_are_independent_and_reset_is_per_stream(weights0):
    from crispy_amd import synth_audio as SA
    T = 20
    x = SA.batch_np(4, T) * np.float32(32768.0)
    ref, _ = _mk(weights0, 4).process(x)
    for b in range(4):      # stream b alone gives the same samples as inside the batch
        solo, _ = _mk(weights0, 1).process(np.ascontiguousarray(x[:, b:b + 1]))
        assert np.array_equal(solo[:, 0], ref[:, b])
    ds = _mk(weights0, 4)
    ds.process(x)
    ds.reset(2)             # audio.rs:955-965: a fresh DenoiseState for one stream only
    again, _ = ds.process(x)
    assert np.array_equal(again[:, 2], ref[:, 2])
    assert not np.array_equal(again[:, 1], ref[:, 1])
    ds.reset(-1)
    again, _ = ds.process(x)
    assert np.array_equal(again, ref)


def test_single_frame_process_frame_semantics(oracle, weights0):
    """DenoiseState::process_frame(out, in) -> vad, one 480-sample frame at a time (audio.rs:268)."""
    from crispy_amd import synth_audio as SA
    x = (SA.stream_np(2, 8) * np.float32(32768.0)).reshape(8, 480)
    ds = _mk(weights0, 1)
    st = oracle.OracleDenoiseState(weights0)
    for t in range(8):
        out = np.empty(480, np.float32)
        vad = ds.process_frame(out, x[t])
        ro, rv = st.process_frame(x[t])
        _assert_pcm_close(out, ro, f"frame {t}")
        assert abs(vad - rv) < 1e-4
    with pytest.raises(ValueError):
        ds.process_frame(np.empty(480, np.float32), x[0][:100])


def test_error_paths_on_device(weights0):
    from crispy_amd import _native as N
    ds = _mk(weights0, 2)
    with pytest.raises(N.CrispyError):
        ds.reset(5)
    with pytest.raises(N.CrispyError):
        _mk(weights0, 2).__class__(weights0, 2, 99)   # device out of range
    assert N.lib().crispy_rn_process(ds._h, None, None, None, 1, 0) == -1
    assert N.lib().crispy_rn_process(ds._h, None, None, None, 0, 0) == 0    # empty input is a no-op


def test_full_size_properties_4096_streams(weights0):
    """BASELINE cfg 2 size (4096 streams), properties that need no oracle run:
    * every 10th stream is digital silence -> output exactly zero, VAD zero;
    * streams fed identical audio produce identical output wherever they sit in the batch;
    * a sample of 16 streams matches solo runs bit for bit; everything is finite and within int16 range."""
    import torch
    from crispy_amd import synth_audio as SA
    B, T = 4096, 25
    dev = torch.device("cuda:0")
    d_in = SA.batch_torch(B, T, dev, seed=3)
    d_in[:, 1000] = d_in[:, 17]
    d_in[:, 4095] = d_in[:, 17]
    d_out = torch.empty_like(d_in)
    d_vad = torch.empty(T, B, device=dev)
    ds = _mk(weights0, B)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad.data_ptr())
    ds.synchronize()
    out, vad = d_out.cpu().numpy(), d_vad.cpu().numpy()
    assert np.isfinite(out).all() and np.abs(out).max() < 40000
    silent = np.arange(B) % 10 == 9
    assert np.all(out[:, silent] == 0.0) and np.all(vad[:, silent] == 0.0)
    assert np.array_equal(out[:, 1000], out[:, 17]) and np.array_equal(out[:, 4095], out[:, 17])
    x = d_in.cpu().numpy()
    pick = [0, 1, 63, 64, 65, 511, 1023, 1024, 2047, 2048, 3000, 3071, 3072, 4000, 4094, 4095]
    solo, _ = _mk(weights0, len(pick)).process(np.ascontiguousarray(x[:, pick]))
    assert np.array_equal(solo, out[:, pick])
    e_in = (x[5:] ** 2).sum()
    assert (out[5:] ** 2).sum() <= 1.05 * e_in       # gains <= 1: the denoiser never adds energy


def test_sampled_oracle_parity_at_full_size(oracle, weights0):
    """4096-stream batch, 40 frames: 24 sampled streams against the oracle -- PCM, and the pitch index held to equality
    outside the oracle's decision margin (the taps form of the frame kernel: same arithmetic, captures on)."""
    import torch
    from crispy_amd import synth_audio as SA
    B, T = 4096, 40
    dev = torch.device("cuda:0")
    d_in = SA.batch_torch(B, T, dev, seed=11)
    d_out = torch.empty_like(d_in)
    d_taps = torch.zeros(T, B, 72, device=dev)
    ds = _mk(weights0, B)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_taps=d_taps.data_ptr())
    ds.synchronize()
    pick = list(range(0, B, 171))
    x = d_in[:, pick].cpu().numpy()
    out = d_out[:, pick].cpu().numpy()
    tp = d_taps[:, pick, 64].cpu().numpy()
    bad = n_diff = n_close = 0
    per_stream = []
    for i, b in enumerate(pick):
        ro, _, rt, margin = oracle.OracleDenoiseState(weights0).process(np.ascontiguousarray(x[:, i]), with_taps=True, with_margin=True)
        err = np.abs(out[:, i] - ro).max()
        if err > REL * np.abs(ro).max() + 1e-3:
            bad += 1
        d, c = _account_for_pitch(tp[:, i], rt[:, 64], margin, f"stream {b}")
        per_stream.append(d)
        n_diff += d
        n_close += c
    assert bad == 0, f"{bad}/{len(pick)} sampled streams out of tolerance"
    n = len(pick) * T
    print(f"pitch index at 4096 streams: {n_diff} of {n} sampled frames differ (per stream {per_stream}); {n_close} frames inside the margin")
    assert n_diff <= 0.002 * n + 1, (n_diff, per_stream)


def test_no_kernel_reads_lds_it_has_not_written(oracle, weights0):
    """LDS is not cleared between workgroups: a kernel that reads a word before writing it sees what the previous kernel
    on that CU left behind, and is right or wrong depending on the history of the process (round 3: a since-retired
    pipeline failed in one pytest process out of three, never alone).  libcrispy_hip_poison.so (`make variants`,
    RN_POISON_LDS) is the same code with every RNNoise kernel filling its LDS allocation with NaNs first: it must still
    match the oracle, with taps (the DBG instantiation) and without."""
    import torch
    from crispy_amd import _native as N
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import DenoiseState
    L = N.load_variant("poison")
    B, T = 9, 41                                       # sub-chunks of 3, 4, 5, 7, 10, 12 frames: groups of 5 + ragged ends
    x = SA.batch_np(B, T, first_stream=300) * np.float32(32768.0)
    x[20:, 4] = 0.0                                    # one stream falls silent (the silence path skips most stages)
    ds = DenoiseState(weights0, B, 0, lib=L)
    o1, v1 = ds.process(x)
    assert np.isfinite(o1).all() and np.isfinite(v1).all(), "NaN: some kernel read LDS it had not written"
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    d_vad = torch.zeros(T, B, device=dev)
    d_taps = torch.zeros(T, B, 72, device=dev)
    torch.cuda.synchronize()
    dt = DenoiseState(weights0, B, 0, lib=L)
    dt.process_device(d_in.data_ptr(), d_out.data_ptr(), T, d_vad.data_ptr(), d_taps.data_ptr())
    dt.synchronize()
    o2, taps = d_out.cpu().numpy(), d_taps.cpu().numpy()
    assert np.isfinite(o2).all() and np.isfinite(taps).all(), "NaN with taps: some kernel read LDS it had not written"
    for b in range(B):
        ro, rv = oracle.OracleDenoiseState(weights0).process(x[:, b])
        _assert_pcm_close(o1[:, b], ro, f"poisoned LDS, stream {b}")
        _assert_pcm_close(o2[:, b], ro, f"poisoned LDS with taps, stream {b}")
        assert np.abs(v1[:, b] - rv).max() < 1e-4


def test_long_run_no_drift(oracle, weights0):
    """10 s of audio (1000 frames, 40 launches, 4 internal 250-frame segments): recurrent state (GRUs, pitch
    continuity, cepstral ring, OLA) must not drift away from the oracle; checked on the LAST second."""
    from crispy_amd import synth_audio as SA
    B, T = 6, 1000
    x = SA.batch_np(B, T, first_stream=40) * np.float32(32768.0)
    ds = _mk(weights0, B)
    out, vad = ds.process(x)
    for b in range(B):
        ro, rv = oracle.OracleDenoiseState(weights0).process(x[:, b])
        peak = max(np.abs(ro).max(), 1.0)
        tail_err = np.abs(out[900:, b] - ro[900:]).max() / peak
        if tail_err > 1e-4:      # where did it leave the oracle?
            e = np.abs(out[:, b] - ro).max(axis=1) / peak
            bad = np.flatnonzero(e > 1e-4)
            w = int(e.argmax())
            es = np.flatnonzero(np.abs(out[w, b] - ro[w]) / peak > 1e-4)
            pytest.fail(f"stream {b}: tail error {tail_err:.3e}; frames above 1e-4: "
                        f"{bad[:24].tolist()} ({bad.size} of {T}), worst {e.max():.3e} at frame {w}, samples {es.min()}..{es.max()} ({es.size}) of it; "
                        f"vad error there {abs(float(vad[int(e.argmax()), b] - rv[int(e.argmax())])):.2e}")
        assert np.abs(vad[900:, b] - rv[900:]).max() < 1e-4


def test_adapter_playback_and_input_resampler(oracle, weights0):
    """RnnNoiseProcessor beyond push_sample (audio.rs:216-240, 297-314): a 44.1 kHz device goes through the input
    LinearResampler to 48 kHz before framing; next_sample() interpolates the output ring at input_rate/output_rate."""
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import LinearResampler, RnnNoiseProcessor
    x48 = SA.stream_np(7, 12, silent=False)
    # (a) 44.1 kHz input: emulate the adapter with the oracle behind the same resampler
    n441 = int(len(x48) * 44100 / 48000)
    x441 = np.interp(np.arange(n441) * (48000 / 44100), np.arange(len(x48)), x48).astype(np.float32)
    proc = RnnNoiseProcessor(weights0, 44100.0, 48000.0, 1.0, 1, 0)
    assert proc.produced_rate_hz() == 48000.0
    got = []
    for s in x441:
        r = proc.push_sample([s])
        if r is not None:
            got.append(r[:, 0])
    got = np.concatenate(got)
    rs = LinearResampler(44100.0, 48000.0)
    ups = []
    for s in x441:
        rs.process_sample(s, ups.append)
    ups = np.array(ups, np.float32)
    nfr = len(ups) // 480
    ref, _ = oracle.OracleDenoiseState(weights0).process(ups[:nfr * 480] * np.float32(32768.0))
    want = np.clip(ref / np.float32(32768.0), -1, 1)[1:].ravel()
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-7
    # (b) playback: output_rate 24 kHz -> every second sample of the ring, linearly interpolated
    proc = RnnNoiseProcessor(weights0, 48000.0, 24000.0, 1.0, 1, 0)
    assert np.all(proc.next_sample() == 0)                      # fewer than two samples buffered
    for s in x48[:480 * 3]:
        proc.push_sample([s])
    ring = np.array([v[0] for v in proc.output_buf])
    assert ring.size == 960                                     # 3 frames pushed, the first one dropped
    play = np.array([proc.next_sample()[0] for _ in range(200)])
    assert np.allclose(play, ring[0:400:2], atol=1e-7)


def test_capture_callback_glue_with_the_denoiser(oracle, weights0):
    """push_mono_to_buffers with an active RnnNoise suppressor (audio.rs:682-730): what lands in the recording ring
    is the adapter's output (48 kHz -> no recording resampling), the level meter sees the raw input."""
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import CaptureBuffers, RnnNoiseProcessor
    x = SA.stream_np(9, 6, silent=False)
    proc = RnnNoiseProcessor(weights0, 48000.0, 48000.0, 1.0, 1, 0)
    cb = CaptureBuffers()
    for s in x:
        cb.push_mono(s, proc, 48000.0)
    ref, _ = oracle.OracleDenoiseState(weights0).process(x * np.float32(32768.0))
    want = np.clip(ref / np.float32(32768.0), -1, 1)[1:].ravel()          # first frame dropped by the adapter
    got = np.array(cb.rec_buffer, np.float32)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-7
    assert abs(cb.rms() - float(np.sqrt(np.mean(x.astype(np.float64) ** 2)))) < 1e-5


# ---- parity hardening: weight extremes through the gain network -------------------------------------
XGOLD = os.path.join(os.path.dirname(__file__), "golden", "rnnoise_extreme_golden.npz")


@pytest.mark.parametrize("kind", ["pos127", "neg127", "alt127", "zero", "bias_pos127", "bias_neg127", "heavy_tail",
                                  "row_saturating"])
def test_weight_extremes_through_the_gain_network(oracle, kind):
    """All-+127, all--127, alternating +-127, zero, saturated biases, a heavy-tailed trained-like draw and rows that
    pin gates at the +-8 clamp: PCM, gains and VAD of the gain network (int8 weights on v_mfma_i32_4x4x4_16B_i8, the
    activations as per-vector power-of-two fixed point) against the oracle -- 40 frames live, and the committed 12-frame
    golden vectors.  (Rounds 1 - 3 kept three more forms of the network alive beside this one -- v_fma_mix_f32, f16 MFMA,
    a stream-batched bf16 MFMA kernel -- and ran them through these cases too; the f16 form overflowed on
    `row_saturating`, which is why the shipped form scales every vector.  They were retired in round 4.)"""
    from crispy_amd import rnn_weights as RW
    from crispy_amd.denoise import DenoiseState
    import torch
    w = RW.extreme_weights(kind)
    G = np.load(XGOLD)
    # (1) golden vectors, one stream per input
    names = ("tone", "loud")
    xg = np.stack([G[f"x/{n}"] for n in names], axis=1)                  # [12, 2, 480]
    ds = DenoiseState(w, 2, 0)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(xg).to(dev)
    d_out = torch.empty_like(d_in)
    d_vad = torch.zeros(12, 2, device=dev)
    d_taps = torch.zeros(12, 2, 72, device=dev)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), 12, d_vad.data_ptr(), d_taps.data_ptr())
    ds.synchronize()
    out, vad, taps = d_out.cpu().numpy(), d_vad.cpu().numpy(), d_taps.cpu().numpy()
    for i, n in enumerate(names):
        _assert_pcm_close(out[:, i], G[f"{kind}/{n}/out"], f"{kind}/{n} golden")
        assert np.abs(vad[:, i] - G[f"{kind}/{n}/vad"]).max() < 1e-4
        assert np.abs(taps[:, i, 42:64] - G[f"{kind}/{n}/gains"]).max() < 1e-4
    # (2) 40 frames live against the oracle, 5 different streams (incl. one that goes silent half way)
    from crispy_amd import synth_audio as SA
    T, B = 40, 5
    x = SA.batch_np(B, T, first_stream=200) * np.float32(32768.0)
    x[T // 2:, 3] = 0.0
    ds = DenoiseState(w, B, 0)
    o2, v2 = ds.process(x)
    assert np.isfinite(o2).all()
    for b in range(B):
        ro, rv = oracle.OracleDenoiseState(w).process(x[:, b])
        _assert_pcm_close(o2[:, b], ro, f"{kind} stream {b}")
        assert np.abs(v2[:, b] - rv).max() < 1e-4, (kind, b)


def test_tansig_and_sigmoid_every_table_cell_and_both_clamps(oracle, weights0):
    """crispy_rn_stage_tansig_device = the frame kernel's activation code (table in registers, ds_bpermute lookups):
    bit-compared with the oracle's tansig_approx / sigmoid_approx on (a) a dense sweep of [-9, 9] (every one of the
    201 cells, 90 points per cell), (b) both sides of every cell boundary (|x| = (i + 0.5) / 25 and its f32
    neighbours), (c) the +-8 clamps and their neighbours, (d) signed zero, subnormals, huge values, infinities, NaN."""
    import torch
    cells = (np.arange(0, 201) + 0.5) / 25.0
    edge = np.concatenate([np.nextafter(cells.astype(np.float32), np.float32(0)), cells.astype(np.float32),
                           np.nextafter(cells.astype(np.float32), np.float32(100))])
    clamp = np.array([8.0, np.nextafter(np.float32(8), np.float32(0)), np.nextafter(np.float32(8), np.float32(9)),
                      16.0, 15.999999, 16.000002], np.float32)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 1e-20, 1e10, -1e10, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan],
                       np.float32)
    sweep = np.linspace(-9.0, 9.0, 90 * 201 * 2 + 1).astype(np.float32)
    xs = np.concatenate([sweep, edge, -edge, 2 * edge, -2 * edge, clamp, -clamp, special]).astype(np.float32)
    ds = _mk(weights0, 1)
    dev = torch.device("cuda:0")
    d_x = torch.from_numpy(xs).to(dev)
    d_y = torch.empty_like(d_x)
    lib = oracle.lib()
    for sigmoid, fn in ((False, lib.rno_tansig_approx), (True, lib.rno_sigmoid_approx)):
        ds.stage_tansig_device(d_x.data_ptr(), d_y.data_ptr(), xs.size, sigmoid)
        ds.synchronize()
        got = d_y.cpu().numpy()
        want = np.array([fn(float(v)) for v in xs], np.float32)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        # +0 / -0 are the same value to every consumer of the activations
        same |= (got == 0) & (want == 0)
        bad = np.flatnonzero(~same)
        assert bad.size == 0, (sigmoid, xs[bad[:8]], got[bad[:8]], want[bad[:8]])
    # the sweep really visits every cell
    idx = np.floor(0.5 + 25.0 * np.minimum(np.abs(sweep), 8.0)).astype(int)
    assert set(idx.tolist()) == set(range(201))


def test_create_from_rnnoise_nu_model_file(oracle, weights0, tmp_path):
    """DenoiseState::new() has no arguments (audio.rs:229); the C ABI's counterpart for a host without a blob of its
    own is crispy_rn_create_from_file over the rnnoise-nu text format: same results as the blob constructor."""
    from crispy_amd import rnn_weights as RW, synth_audio as SA
    from crispy_amd.denoise import DenoiseState
    p = tmp_path / "model.txt"
    RW.save_rnnoise_nu_text(str(p), weights0)
    x = SA.batch_np(3, 15, first_stream=70) * np.float32(32768.0)
    a, va = DenoiseState(str(p), 3, 0).process(x)
    b, vb = DenoiseState(weights0, 3, 0).process(x)
    assert np.array_equal(a, b) and np.array_equal(va, vb)


# ---- BASELINE configs at full size ----------------------------------------------------------------
def test_cfg1_full_length_clip_3000_frames_through_the_adapter(oracle, weights0):
    """BASELINE configs[0] at its full size: one 30 s 48 kHz mono clip = 3000 frames pushed SAMPLE BY SAMPLE through
    the RnnNoiseProcessor adapter (push_sample: x32768, process_frame per 480 samples, /32768, clamp, volume, first
    frame dropped -- audio.rs:242-295) over the HIP path, against the same adapter arithmetic over the oracle's
    process_frame loop."""
    from crispy_amd import synth_audio as SA
    from crispy_amd.denoise import RnnNoiseProcessor
    T = 3000
    clip = SA.cfg1_clip(T)
    assert clip.size == 1_440_000
    proc = RnnNoiseProcessor(weights0, 48000.0, 48000.0, 0.8, 1, 0)
    chunks = []
    for s in clip:
        r = proc.push_sample([s])
        if r is not None:
            chunks.append(r[:, 0])
    got = np.concatenate(chunks)
    ro, _ = oracle.OracleDenoiseState(weights0).process(clip * np.float32(32768.0))
    want = (np.clip(ro / np.float32(32768.0), -1, 1) * np.float32(0.8))[1:].ravel()
    assert got.shape == want.shape == ((T - 1) * 480,)
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-7
    # the last second on its own (state after 29 s)
    assert np.abs(got[-48000:] - want[-48000:]).max() <= 1e-4 * np.abs(want[-48000:]).max() + 1e-7


def test_cfg2_full_size_4096_streams_x_1000_frames(oracle, weights0):
    """BASELINE configs[1] at its full size: 4096 concurrent streams x 1000 frames (10 s) in one device-resident call;
    24 sampled streams against the oracle on the LAST second (recurrent state after 900 frames), silent streams
    exactly zero, everything finite."""
    import torch
    from crispy_amd import synth_audio as SA
    B, T = 4096, 1000
    dev = torch.device("cuda:0")
    d_in = SA.batch_torch(B, T, dev, seed=21)                           # 7.9 GB
    d_out = torch.empty_like(d_in)
    ds = _mk(weights0, B)
    torch.cuda.synchronize()
    ds.process_device(d_in.data_ptr(), d_out.data_ptr(), T)
    ds.synchronize()
    assert bool(torch.isfinite(d_out).all())
    silent = torch.arange(B, device=dev) % 10 == 9
    assert float(d_out[:, silent].abs().max()) == 0.0
    pick = list(range(5, B, 171))
    assert len(pick) == 24
    x = d_in[:, pick].cpu().numpy()
    out = d_out[:, pick].cpu().numpy()
    del d_in, d_out
    torch.cuda.empty_cache()
    for i, b in enumerate(pick):
        ro, _ = oracle.OracleDenoiseState(weights0).process(np.ascontiguousarray(x[:, i]))
        peak = max(float(np.abs(ro).max()), 1.0)
        err = float(np.abs(out[900:, i] - ro[900:]).max())
        assert err <= REL * peak + 1e-3, (b, err, peak)
