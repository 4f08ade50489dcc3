"""The header itself, not ctypes prototypes: a C99 program compiled against include/crispy_hip.h drives the
single-stream `process_frame` drop-in (one 480-sample frame per call from host slices, audio.rs:260-268) and its
output is compared with the oracle; the same program reports the per-call latency bench.py publishes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_compiles_against_the_header_as_c99(tmp_path):
    """CPU: the program builds with -std=c99 -pedantic -Werror and, without a device, fails loudly through the ABI's
    own status/message path (no CPU fallback)."""
    from tests import c_dropin
    exe = c_dropin.build(str(tmp_path))
    from crispy_amd import _native as N
    if N.lib().crispy_device_count() > 0:
        pytest.skip("a GPU is present: the device path is covered by the gpu test below")
    from crispy_amd import rnn_weights as RW
    model = tmp_path / "m.txt"
    RW.save_rnnoise_nu_text(str(model), RW.synthetic_weights(0))
    (tmp_path / "in.f32").write_bytes(np.zeros(480, np.float32).tobytes())
    r = subprocess.run([exe, str(model), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), "1", "0"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_dropin_matches_oracle_and_reports_latency(oracle, weights0):
    from crispy_amd import synth_audio as SA
    from tests import c_dropin
    T = 40
    x = (SA.stream_np(11, T, silent=False) * np.float32(32768.0)).reshape(T, 480)
    out, vad, lat = c_dropin.run(weights0, x, timed_calls=300)
    ro, rv = oracle.OracleDenoiseState(weights0).process(x)
    assert np.abs(out - ro).max() <= 1e-4 * np.abs(ro).max() + 1e-3
    assert np.abs(vad - rv).max() < 1e-4
    assert lat["calls"] == 300 and 0 < lat["p50"] <= lat["p99"] <= lat["max"]
    assert lat["p99"] < 10000, f"a process_frame call must fit the 10 ms audio callback budget: {lat}"


def _build_multi_gpu(out_dir):
    exe = os.path.join(out_dir, "multi_gpu")
    lib_dir = os.path.join(ROOT, "crispy_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-D_POSIX_C_SOURCE=199309L", "-pthread",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "multi_gpu.c"), "-o", exe,
                    "-L", lib_dir, "-lcrispy_hip", f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True, text=True)
    return exe


def test_multi_gpu_c_program_compiles_as_c99(tmp_path):
    """CPU: tests/c/multi_gpu.c builds against the header with -std=c99 -pedantic -Werror and, without a device, says so."""
    exe = _build_multi_gpu(str(tmp_path))
    from crispy_amd import _native as N
    if N.lib().crispy_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test below")
    r = subprocess.run([exe, "m", "i", "o", "4", "2", "2"], capture_output=True, text=True)
    assert r.returncode == 3 and "no gfx950 device" in r.stderr


@pytest.mark.gpu
def test_multi_gpu_c_program_shards_by_stream_id_without_changing_a_sample(oracle, weights0, tmp_path):
    """The stream-sharded split from the C ABI alone (no Python, no torch.distributed): one process, one handle and one
    host thread per shard, block partition by stream id, shard r on device r % crispy_device_count().  The shards are
    independent, so 1, 3 and 8 shards give the same bytes and the same checksum, equal to the oracle's per stream.  On
    a box with several GPUs the shards really sit on different devices; on a 1-GPU box they share the one there is
    (same code path: per-shard handles, threads, partition)."""
    import json
    from crispy_amd import rnn_weights as RW, synth_audio as SA
    exe = _build_multi_gpu(str(tmp_path))
    B, T = 24, 12
    x = SA.batch_np(B, T, first_stream=700) * np.float32(32768.0)                  # [T, B, 480]
    model = tmp_path / "m.txt"
    RW.save_rnnoise_nu_text(str(model), weights0)
    (tmp_path / "in.f32").write_bytes(np.ascontiguousarray(x).tobytes())
    outs, sums = {}, {}
    for shards in (1, 3, 8):
        fo = tmp_path / f"out{shards}.f32"
        r = subprocess.run([exe, str(model), str(tmp_path / "in.f32"), str(fo), str(B), str(T), str(shards)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["shards"] == shards and info["devices"] >= 1
        outs[shards] = np.fromfile(fo, dtype=np.float32).reshape(T, B, 480)
        sums[shards] = info["checksum"]
    assert np.array_equal(outs[1], outs[3]) and np.array_equal(outs[1], outs[8])
    assert sums[1] == sums[3] == sums[8]
    for b in (0, 7, 23):
        ro, _ = oracle.OracleDenoiseState(weights0).process(np.ascontiguousarray(x[:, b]))
        assert np.abs(outs[8][:, b] - ro).max() <= 1e-4 * np.abs(ro).max() + 1e-3


def _build_c(out_dir, name):
    exe = os.path.join(out_dir, name)
    lib_dir = os.path.join(ROOT, "crispy_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-D_POSIX_C_SOURCE=199309L", "-pthread",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe,
                    "-L", lib_dir, "-lcrispy_hip", f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True, text=True)
    return exe


def test_multi_gpu_asr_c_program_compiles_as_c99(tmp_path):
    """CPU: tests/c/multi_gpu_asr.c (cfg 5's split from the C ABI) builds with -std=c99 -pedantic -Werror, checks the ABI
    version the header declares against the library's, and without a device says so."""
    exe = _build_c(str(tmp_path), "multi_gpu_asr")
    from crispy_amd import _native as N
    if N.lib().crispy_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test below")
    r = subprocess.run([exe, "m", "p", "4", "16000", "2", "4"], capture_output=True, text=True)
    assert r.returncode == 3 and "no gfx950 device" in r.stderr


@pytest.mark.gpu
def test_multi_gpu_asr_c_program_deals_chunks_in_blocks_without_changing_a_token(tmp_path):
    """BASELINE configs[4]'s split from the C ABI alone: one crispy_asr engine + one host thread per shard in one process,
    30 s chunks dealt in blocks by shard_range, one crispy_asr_transcribe_batch per shard (timestamps on, seek loop, one
    greedy pass per window).  1 and 3 shards give the same ids per chunk, equal to the Python binding's run on ONE
    engine; on a 1-GPU box all shards share the device (same code path), on a node each sits on its own."""
    import json
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.sharding import shard_range
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0, sensitive=True)
    model = tmp_path / "tiny.bin"
    write_ggml(str(model), hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), f16=True)
    N, S, max_new = 7, 16000 * 12, 8
    pcm = np.stack([synth_audio.clip16k_np(900 + i, S) for i in range(N)])
    (tmp_path / "pcm.f32").write_bytes(np.ascontiguousarray(pcm, dtype=np.float32).tobytes())
    exe = _build_c(str(tmp_path), "multi_gpu_asr")
    runs = {}
    for shards in (1, 3):
        r = subprocess.run([exe, str(model), str(tmp_path / "pcm.f32"), str(N), str(S), str(shards), str(max_new)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["shards"] == shards and len(info["chunks"]) == N
        runs[shards] = info["chunks"]
    assert runs[1] == runs[3]
    assert [shard_range(N, r, 3) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]        # the blocks the C program deals (same formula)
    eng = WhisperEngine(str(model), resident=True)
    eng.set_precision(1)
    ref = transcribe_batch(eng, list(pcm), max_new_tokens=max_new, timestamps=True, fallback=False)
    eng.close()
    assert [c["tokens"] for c in runs[3]] == [toks for _, toks, _ in ref]
    assert [c["lang"] for c in runs[3]] == [lang for _, _, lang in ref]
    assert len({tuple(c["tokens"]) for c in runs[3]}) >= 4                                # the chunks do decode differently
