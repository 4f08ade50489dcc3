"""The library's device-free HOST logic under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU box (VERDICT r4
next #6).  The library parses untrusted model files -- whisper.cpp GGML containers, rnnoise-nu text models -- inside a host
built with `panic = "abort"` (src-tauri/Cargo.toml:10-20: there is nothing to catch a crash), and the corrupt-file tests of
tests/test_boundary_robustness.py check status codes, not memory safety.  Here the product's own translation units
(api_util.cpp, asr_api.cpp, crispy_api.cpp, whisper_api.cpp) are compiled unmodified by tests/asan/harness.cpp against a
host-memory stand-in for the HIP runtime (tests/asan/hip/hip_runtime.h: "device" buffers are malloc'd, so every copy the
loaders make is bounds-checked; kernel launchers are no-ops), with -fsanitize=address,undefined, and fed:

  * the hand-made corrupt GGML files of tests/test_boundary_robustness.py (absurd dims, overflowing shapes, unknown / wrong-
    sized tensors) -- the cases that on a CPU-only box stop at NO_DEVICE before the reader sees them;
  * >= 2 000 seeded mutations (truncations, bit flips in the header / tensor headers / names, hostile 32-bit fields, hostile
    vocabulary lengths) of a valid one-layer Whisper file, through both loaders (inflate at load; resident quantised);
  * >= 2 000 seeded mutations of a valid rnnoise-nu text model;
  * whisper_full's decision logic -- replay_decoder / score_decoder / window_segments and the std::mt19937 variate -- on
    sequences produced by the oracle's decode_temperature over scripted decoders, compared field by field (a CPU parity test
    of the C++ the GPU tests otherwise only reach through a device).

Done = zero sanitizer reports, every mutation answered with a crispy_status code.  CPU build only: GPU sanitizers are not
available on the pool and are not used."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
N_MUT = int(os.environ.get("CRISPY_FUZZ_MUTATIONS", "2000"))

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ (sanitizer runtimes, _Float16) not installed")


@pytest.fixture(scope="session")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("asan") / "harness")
    r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "tests", "asan"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "asan", "harness.cpp"), "-o", exe, "-lpthread"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, args, stdin=None, timeout=900):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, *args], input=stdin, capture_output=True, text=True, timeout=timeout, env=env)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, \
        (args, r.stderr[-4000:])
    return r


@pytest.fixture(scope="session")
def one_layer_files(tmp_path_factory):
    """A valid one-layer Whisper-tiny-width model (full vocabulary: the loader insists on a Whisper vocabulary size) as an
    f16 file and as a q5_0 file, plus the byte offsets of every tensor header and vocabulary length field."""
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml, write_ggml_quantized
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams(n_audio_layer=1, n_text_layer=1)
    W = synthetic_whisper_weights(hp, 2)
    # the 20 M-element token embedding LAST in the file (the format fixes no order): a mutated header of any other tensor is
    # then met after 3 MB of reading instead of 43 MB -- the mutations test the reader, not fread
    emb = "decoder.token_embedding.weight"
    W = {**{k: v for k, v in W.items() if k != emb}, emb: W[emb]}
    d = tmp_path_factory.mktemp("ggml1")
    files = {}
    for kind in ("f16", "q5_0"):
        p = str(d / f"one-layer-{kind}.bin")
        if kind == "f16":
            write_ggml(p, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), f16=True)
        else:
            write_ggml_quantized(p, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), kind, keep=False)
        files[kind] = (p, _offsets(p, str(d / f"offsets-{kind}.txt")))
    return hp, files


BLOCK_BYTES = {0: (1, 4), 1: (1, 2), 2: (32, 18), 3: (32, 20), 6: (32, 22), 7: (32, 24), 8: (32, 34)}      # ggml type -> (elements, bytes) per block


def _offsets(path, out):
    """The file's structure, read by this test's own parser: tensor header offsets, a -1, the
    offsets of the vocabulary's length fields."""
    b = open(path, "rb").read()
    o = 4 + 11 * 4
    n_mel, n_fft = struct.unpack_from("<2i", b, o)
    o += 8 + 4 * n_mel * n_fft
    (n_tok,) = struct.unpack_from("<i", b, o)
    o += 4
    voc = []
    for _ in range(n_tok):
        voc.append(o)
        (ln,) = struct.unpack_from("<I", b, o)
        o += 4 + ln
    tens = []
    while o < len(b):
        tens.append(o)
        nd, nl, tt = struct.unpack_from("<3i", b, o)
        dims = struct.unpack_from(f"<{nd}i", b, o + 12)
        n = int(np.prod(dims))
        el, by = BLOCK_BYTES[tt]
        o += 12 + 4 * nd + nl + n // el * by
    assert o == len(b)
    with open(out, "w") as f:
        f.write("\n".join(map(str, tens)) + "\n-1\n" + "\n".join(map(str, voc[::97])) + "\n")      # every 97th token: 535 of them
    return out


def test_valid_files_load_and_hand_made_corrupt_ones_are_rejected(harness, one_layer_files, tmp_path):
    hp, files = one_layer_files
    for kind, (path, _) in files.items():
        for mode in ([], ["resident"]):
            r = _run(harness, ["load", path, *mode])
            assert r.returncode == 0 and json.loads(r.stdout)["status"] == 0, (kind, mode, r.stdout, r.stderr[-500:])
    # tests/test_boundary_robustness.py's corrupt files: here the reader really sees them (a CPU box stops at NO_DEVICE there)
    from tests.test_boundary_robustness import _ggml_header, _tensor
    cases = {
        "bad magic": b"abcd",
        "n_vocab": _ggml_header(n_vocab=2 ** 31 - 1),
        "": _ggml_header(vocab_entries=-3),
        "overflows": _ggml_header() + _tensor("encoder.conv1.bias", [2 ** 30, 2 ** 30, 4]),
        "unknown tensor": _ggml_header() + _tensor("encoder.bogus", [4]),
        "imply 384": _ggml_header() + _tensor("encoder.conv1.bias", [2 ** 30]),
        "truncated": _ggml_header() + _tensor("encoder.conv1.bias", [384, 1], data=b"\0" * 100),
    }
    for word, blob in cases.items():
        p = tmp_path / "m.bin"
        p.write_bytes(blob)
        for mode in ([], ["resident"]):
            r = _run(harness, ["load", str(p), *mode])
            out = json.loads(r.stdout)
            assert r.returncode == 1 and out["status"] == -5 and word in out["error"], (word, mode, out)


@pytest.mark.parametrize("kind,resident", [("f16", False), ("q5_0", False), ("q5_0", True)])
def test_seeded_mutations_of_a_valid_ggml_file_only_ever_produce_status_codes(harness, one_layer_files, tmp_path, kind, resident):
    hp, files = one_layer_files
    path, offsets = files[kind]
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    work = os.path.join(scratch, f"crispy-fuzz-{os.getpid()}-{kind}-{int(resident)}.bin")
    shutil.copyfile(path, work)
    try:
        n = N_MUT // 3 + 1                                   # three loader configurations share the count: 2 001 in all
        r = _run(harness, ["fuzz-ggml", work, str(n), str(17 + int(resident)), offsets, *(["resident"] if resident else [])], timeout=1500)
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["mutations"] == n
        by = np.array(out["by_kind"])
        assert by.sum() == n and by[:, 2].sum() <= n // 50, out      # "other" statuses: out-of-memory on a hostile size, nothing else
        assert by[:, 1].sum() >= n // 2, out                          # most mutations are caught as a bad model
        assert by[:, 0].sum() >= 1, out                               # and some are harmless (a flipped bit in a name's padding, a payload)
        print(kind, "resident" if resident else "inflate", out["by_kind"])
    finally:
        os.unlink(work)


def test_seeded_mutations_of_an_rnnoise_nu_model(harness, tmp_path):
    from crispy_amd import rnn_weights as RW
    p = tmp_path / "model.txt"
    RW.save_rnnoise_nu_text(str(p), RW.synthetic_weights(0))
    r = _run(harness, ["rnnoise", str(p)])
    assert r.returncode == 0 and json.loads(r.stdout)["status"] == 0
    r = _run(harness, ["fuzz-rnnoise", str(p), str(N_MUT), "5"])
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["mutations"] == N_MUT and out["rejected"] >= N_MUT // 2 and out["parsed"] >= 1, out
    print(out)


def test_decision_logic_of_the_library_equals_the_oracles_on_scripted_decoders(harness):
    """whisper_full's per-pass bookkeeping, the sequence score and the segments, C++ (crispy_amd/csrc/whisper_api.cpp:
    replay_decoder, score_decoder, window_segments) against the oracle (oracle/whisper_oracle.py: decode_temperature,
    sequence_score, window_segments) on 400 scripted passes that reach every branch: sequences that close on a timestamp
    pair, end on EOT with and without a timestamp, go back in time, run into the token limit, repeat themselves, end near
    the end of the audio; greedy and sampled."""
    from crispy_amd.whisper_weights import HParams
    from oracle import whisper_oracle as WO
    from tests.test_oracle_whisper_full import SP, SUP, SUP_FIRST, Scripted, peaky
    hp = HParams.tiny()
    BEG, EOT = SP["beg"], SP["eot"]
    rng = np.random.default_rng(99)
    params = dict(delta_min=10, max_initial_ts=50, length_penalty=-1.0, entropy_thold=2.4)
    init = [SP["sot"], SP["lang0"], SP["transcribe"]]
    lines, refs = [], []
    for case in range(400):
        n_max = int(rng.integers(4, 60))
        seek = int(rng.choice([0, 600, 1400, 2500]))
        seek_end = int(seek + rng.choice([200, 900, 3000, 3000]))
        # a script: timestamps and words, sometimes repeating, sometimes going backwards, sometimes never ending
        seq, ts = [], int(rng.integers(0, 20))
        seq.append(BEG + ts)
        for _ in range(int(rng.integers(1, 70))):
            u = rng.random()
            if u < 0.62:
                seq.append(int(rng.integers(300, 3000)) if rng.random() < 0.7 else 777)
            else:
                ts = max(0, ts + int(rng.integers(-20, 200)))
                seq += [BEG + min(ts, 1500)] * int(rng.integers(1, 3))
        if rng.random() < 0.5:
            seq.append(EOT)
        height = float(rng.choice([30.0, 3.0]))
        t_cur = float(rng.choice([0.0, 0.0, 0.4, 1.0]))
        n_dec = 1 if t_cur == 0.0 else 3
        script = (lambda s, hgt: (lambda gen, prompt: peaky(s[len(gen)] if len(gen) < len(s) else EOT, hgt)))(seq, height)
        out = WO.decode_temperature(Scripted(script), init, SP, WO.RULES_WCPP, n_max, seek, seek_end, t_cur, n_dec,
                                    [WO.MT19937(j) for j in range(n_dec)], params, SUP, SUP_FIRST)
        for d in out["decoders"]:
            n = len(d["toks"])
            lines.append("P %d %d %d %d %d %d %d %s %s %s" % (n_max, BEG, EOT, seek, seek_end, 10, n, " ".join(map(str, d["toks"])),
                                                             " ".join(map(str, d["tids"])), " ".join(repr(float(np.float32(p))) for p in d["plogs"])))
            refs.append((d, seek))
    lines += ["U %d 6" % s for s in (0, 1, 4, 12345)]
    # the ids suppress_nst masks, on a vocabulary that holds some of whisper.cpp's non-speech strings (and near misses)
    vocab = [b" w%d" % i for i in range(300)]
    rs = np.random.default_rng(3)
    cands = [t.encode("utf-8") for t in WO.NON_SPEECH_TOKENS] + [(" " + t).encode("utf-8") for t in WO.NON_SPEECH_TOKENS] + \
            [b" -", b" '", b"-", b"'", b" (x", b"((((", b"", b" "]
    for i, c in zip(rs.permutation(300)[:len(cands)], cands):
        vocab[int(i)] = c
    vocab[299] = b"("                                          # a duplicate string: the first id counts
    import tempfile
    vf = tempfile.NamedTemporaryFile("w", suffix=".hex", delete=False)
    vf.write("".join(t.hex() + "\n" for t in vocab))
    vf.close()
    lines.append("N " + vf.name)
    r = _run(harness, ["decide"], stdin="\n".join(lines) + "\n")
    os.unlink(vf.name)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [json.loads(ln) for ln in r.stdout.strip().splitlines()]
    nst = got.pop()
    assert nst == WO.non_speech_token_ids(vocab) and len(nst) >= 2 * len(WO.NON_SPEECH_TOKENS) - 2, (nst, WO.non_speech_token_ids(vocab))
    assert len(got) == len(refs) + 4
    seen = dict(failed=0, completed=0, limit=0, segs=0)
    for g, (d, seek) in zip(got, refs):
        # the oracle's entropy check happens AFTER scoring (in the ranking), as in the library's caller: compare the pass itself
        failed_before_ranking = d["failed"] and d["score"] is None and not d.get("kept")
        ref_failed = d["failed"] and d["score"] is None
        assert g["failed"] == int(ref_failed), (g, d["toks"], d["failed"], d["score"])
        if ref_failed:
            seen["failed"] += 1
            continue
        assert g["completed"] == int(d["completed"]) and g["result_len"] == d["result_len"] and g["seek_delta"] == d["seek_delta"], (g, d)
        seen["completed"] += g["completed"]
        seen["limit"] += 1 - g["completed"]
        if d["result_len"] > 0:
            sc = WO.sequence_score(d["toks"], [float(np.float32(p)) for p in d["plogs"]], d["result_len"])
            assert g["scored"] == 1
            assert abs(g["sum"] - sc["sum_logprobs"]) <= 1e-4 * max(1.0, abs(sc["sum_logprobs"])), (g, sc)
            assert abs(g["avg"] - sc["avg_logprobs"]) <= 1e-4 * max(1.0, abs(sc["avg_logprobs"])) and abs(g["entropy"] - sc["entropy"]) <= 1e-9
            win = dict(tokens=d["toks"], tids=d["tids"], result_len=d["result_len"], seek_delta=d["seek_delta"])
            segs = WO.window_segments(win, seek, SP, lambda t: b" w%d" % t)
            assert g["segments"] == [[a, b, s.decode()] for a, b, s in segs], (g["segments"], segs)
            seen["segs"] += len(segs)
    assert seen["failed"] >= 20 and seen["completed"] >= 50 and seen["limit"] >= 5 and seen["segs"] >= 100, seen
    for g, s in zip(got[-4:], (0, 1, 4, 12345)):             # std::mt19937 + generate_canonical<double, 53> against the restated generator
        m = WO.MT19937(s)
        assert g == [m.canonical() for _ in range(6)], s
    print(seen)
