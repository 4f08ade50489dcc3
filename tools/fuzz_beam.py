"""Developer tool (GPU box): beam search, product against oracle, on randomly scripted models -- positions with two- and
three-way near ties (so the beams split, merge again and drop out at different steps), beam sizes 2 .. 5, one window per
clip.  Every case compares tokens, the winning decoder and the window statistics of `crispy_asr_transcribe` with the
oracle's whisper_full (decode_temperature(beam_size=)); cases whose draws the oracle itself cannot resolve (a variate
within 1e-4 of an interval end) are skipped and counted.  CASES=24 SEED=0."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperEngine
from crispy_amd.ggml_io import synthetic_vocab, write_ggml
from crispy_amd.mel_filters import whisper_mel_filters
from crispy_amd.whisper_weights import HParams
from oracle import whisper_oracle as WO
from tests.scripted_model import script_rows, scripted_whisper_weights

hp = HParams.tiny()
sp = WO.special_tokens(hp.n_vocab)
BEG, EOT = sp["beg"], sp["eot"]
sup = [sp["sot"], sp["nosp"], sp["translate"], sp["transcribe"], sp["prev"], sp["solm"]] + list(range(sp["lang0"], sp["lang0"] + 99))
sup_first = [220, EOT]
F = whisper_mel_filters(80)
enc0 = np.zeros((hp.n_audio_ctx, hp.n_audio_state))
x = synth_audio.clip16k_np(80, 16000 * 6)
tmp = tempfile.mkdtemp()
n_cases = int(os.environ.get("CASES", 24))
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
ok = skipped = 0
for case in range(n_cases):
    n_text = int(rng.integers(6, 14))
    toks = rng.choice(np.arange(1000, 40000), size=3 * n_text, replace=False).tolist()
    seq = [BEG]
    ties = sorted(rng.choice(np.arange(n_text), size=int(rng.integers(2, 5)), replace=False).tolist())
    for i in range(n_text):
        if i in ties:
            k = int(rng.integers(2, 4))
            w = [1.0] + [1.0 - float(rng.uniform(0.3, 2.5)) * np.sqrt(2.0) / hp.n_text_state for _ in range(k - 1)]
            seq.append([(toks[3 * i + j], w[j]) for j in range(k)])
        else:
            seq.append(toks[3 * i])
    ts = BEG + int(rng.integers(50, 250))
    seq += [ts, ts, EOT]
    W = scripted_whisper_weights(hp, script_rows(2, seq), gain=100.0)
    path = os.path.join(tmp, f"m{case}.bin")
    write_ggml(path, hp, W, F, synthetic_vocab(hp.n_vocab), f16=False)
    eng = WhisperEngine(path)
    beam = int(rng.integers(2, 6))
    best_of = int(rng.integers(2, 6))
    text, segs, got = eng.transcribe_segments(x, language_token=sp["lang0"], beam_size=beam, best_of=best_of)
    wins = eng.last_windows
    rsegs, rkept, rwins = WO.transcribe_timestamps(W, hp, lambda seek: None, x.size, [sp["sot"], sp["lang0"], sp["transcribe"]],
                                                   WO.RULES_WCPP, eng.token_text, suppress=sup, suppress_first=sup_first,
                                                   fallback=True, encoder=lambda mel: enc0, params=dict(beam_size=beam, best_of=best_of))
    eng.close()
    margin = min(min(d["margins"]) for w in rwins for it in w["iterations"] for d in it["decoders"] if d["margins"])
    if margin < 1e-4:
        skipped += 1
        continue
    want = [t for t in rkept if t != EOT]
    assert got == want, (case, beam, got, want)
    assert len(wins) == len(rwins)
    for g, w in zip(wins, rwins):
        assert g["decoder"] == w["decoder"] and abs(g["temperature"] - w["temperature"]) < 1e-6, (case, g, w["decoder"], w["temperature"])
        assert g["failed"] == int(w["failed"]) and g["n_tokens"] == len(w["tokens"]), (case, g, w["tokens"])
        if np.isfinite(w["avg_logprob"]):
            assert abs(g["avg_logprob"] - w["avg_logprob"]) < 1e-4, (case, g, w["avg_logprob"])
    n_split = len({tuple(d["toks"]) for d in rwins[0]["iterations"][0]["decoders"]})
    ok += 1
    print(f"case {case}: beam {beam}, best_of {best_of}, {len(ties)} ties, {n_split} distinct beams at the end of the first pass, "
          f"temperatures {[round(w['temperature'], 1) for w in rwins]}: equal", flush=True)
print(f"{ok} cases equal, {skipped} skipped (oracle margin below 1e-4)")
