"""Pinning kit, step 3 of 3: compare the CPU oracle and the HIP path with vectors dumped from the
REFERENCE's own crates (nnnoiseless 0.5.2, rubato 0.16.2, whisper.cpp through whisper-rs 0.16.0) by
`bindings/rust/crispy-hip-sys/examples/dump_vectors.rs` over the inputs of `tools/make_ref_inputs.py`.

Runs iff CRISPY_REF_VECTORS names that directory; skipped otherwise -- nothing in this repository's build environment
can produce the vectors (no Rust toolchain, crates not vendored: SURVEY.md 8c), so until a maintainer supplies them the
oracle stays "parity unpinned" and these tests document exactly what would be compared and at which tolerance:
denoised PCM within 1e-4 of the clip's peak and the VAD within 1e-4 (north_star), the resampler within 1e-4 of the
peak, Whisper greedy token ids and text identical.  CRISPY_REF_GGML names the model file dump_vectors was given."""
import json
import os

import numpy as np
import pytest

DIR = os.environ.get("CRISPY_REF_VECTORS")
pytestmark = pytest.mark.skipif(not DIR, reason="CRISPY_REF_VECTORS not set: no vectors from the reference's crates to compare with")


def _man():
    with open(os.path.join(DIR, "manifest.json")) as f:
        return json.load(f)


def _f32(name):
    return np.fromfile(os.path.join(DIR, name), dtype="<f4")


def _rn_cases():
    if not DIR:
        return []
    return [c for c in _man()["rnnoise"] if os.path.exists(os.path.join(DIR, c["ref_out"]))]


def _check_rn(out, vad, case):
    ref, rvad = _f32(case["ref_out"]).reshape(-1, 480), _f32(case["ref_vad"])
    peak = max(float(np.abs(ref).max()), 1.0)
    assert np.abs(out - ref).max() <= 1e-4 * peak, (case["name"], np.abs(out - ref).max() / peak)
    assert np.abs(vad - rvad).max() <= 1e-4, case["name"]


@pytest.mark.parametrize("case", _rn_cases(), ids=lambda c: c["name"])
def test_oracle_process_frame_matches_nnnoiseless(oracle, case):
    """oracle/rnnoise_oracle.c against nnnoiseless::DenoiseState::process_frame (audio.rs:268)."""
    from crispy_amd.rnn_weights import load_rnnoise_nu_text
    w = load_rnnoise_nu_text(os.path.join(DIR, case["model"]))
    x = _f32(case["in"]).reshape(-1, 480)
    out, vad = oracle.OracleDenoiseState(w).process(x)
    _check_rn(out, vad, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _rn_cases(), ids=lambda c: c["name"])
def test_hip_process_frame_matches_nnnoiseless(case):
    """The HIP path through the C ABI (crispy_rn_create_from_file + crispy_rn_process) against the same vectors."""
    from crispy_amd.denoise import DenoiseState
    x = _f32(case["in"]).reshape(-1, 1, 480)
    ds = DenoiseState(os.path.join(DIR, case["model"]), 1, 0)
    out, vad = ds.process(x)
    _check_rn(out[:, 0], vad[:, 0], case)


def test_oracle_resampler_matches_rubato():
    """oracle/resample_oracle.py against rubato::FftFixedIn(48000, 16000, 1024, 1, 1) fed in 1024-sample calls
    (commands/transcription.rs:198-208, 314-357) -- settles the buffering question NOTEBOOK.md section 2 leaves open."""
    from oracle import resample_oracle as RO
    r = _man()["resampler"]
    if not os.path.exists(os.path.join(DIR, r["ref_out"])):
        pytest.skip("no resampler vector")
    ref = _f32(r["ref_out"])
    got = RO.resample_48k_to_16k(_f32(r["in"]))
    assert got.size == ref.size, (got.size, ref.size)
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.gpu
def test_hip_transcribe_matches_whisper_cpp():
    """crispy_asr_transcribe (precision mode 1 = ggml's arithmetic, opts = NULL = TranscribeOptions::default()) against
    whisper.cpp's greedy ids, segments and text on the supplied model file (managers/transcription.rs:183-185)."""
    import ctypes as C
    from crispy_amd import _native as N
    from crispy_amd.asr import WhisperEngine, _read_result
    a = _man()["asr"]
    ggml = os.environ.get("CRISPY_REF_GGML")
    if not ggml or not os.path.exists(os.path.join(DIR, a["ref"])):
        pytest.skip("no ASR vector / CRISPY_REF_GGML not set")
    with open(os.path.join(DIR, a["ref"])) as f:
        ref = json.load(f)
    eng = WhisperEngine(ggml)
    eng.set_precision(1)
    x = _f32(a["in"])
    res = C.c_void_p()
    N.check(N.lib().crispy_asr_transcribe(eng._h, x.ctypes.data, x.size, None, C.byref(res)))
    try:
        text, tokens, lang, segs, _wins = _read_result(res)
    finally:
        N.lib().crispy_asr_free_result(res)
    sp = N.vocab_specials(eng.hp.n_vocab)
    ref_text_tokens = [t for t in ref["tokens"] if t < sp.eot]
    assert [t for t in tokens if t < sp.eot] == ref_text_tokens
    assert text == ref["text"]
    assert [(round(s0 * 100), round(s1 * 100)) for s0, s1, _ in segs] == [(s["t0"], s["t1"]) for s in ref["segments"]]
    if ref.get("lang_id", -1) >= 0:
        assert lang == sp.lang0 + ref["lang_id"]
