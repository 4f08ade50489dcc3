"""Developer tool (GPU box): find synthetic clips on which whisper.cpp's previous-text conditioning changes the transcript
(window >= 2 starts with more than 5 s of audio left and its prompt carries the previous window's tokens), then compare
the product with the oracle's seek loop there.  SEEDS=60:90 SECONDS=12 NMAX=10 SENS=0|1"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperEngine
from crispy_amd.ggml_io import synthetic_vocab, write_ggml
from crispy_amd.mel_filters import whisper_mel_filters
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
from oracle import whisper_oracle as WO
hp = HParams.tiny()
W = synthetic_whisper_weights(hp, 0, sensitive=bool(int(os.environ.get("SENS", "0"))))
F = whisper_mel_filters(80)
path = os.path.join(tempfile.mkdtemp(), "m.bin")
write_ggml(path, hp, W, F, synthetic_vocab(hp.n_vocab), f16=False)
eng = WhisperEngine(path)
sp = WO.special_tokens(hp.n_vocab)
sup = sorted([sp["sot"], sp["nosp"], sp["translate"], sp["transcribe"], sp["prev"], sp["solm"]] + list(range(sp["lang0"], sp["lang0"] + sp["n_lang"])))
lo, hi = [int(v) for v in os.environ.get("SEEDS", "60:80").split(":")]
secs = float(os.environ.get("SECONDS", "12")); nmax = int(os.environ.get("NMAX", "10"))
from tests import oracle_lib
oracle_lib.lib()
for seed in range(lo, hi):
    x = synth_audio.clip16k_np(seed, int(16000 * secs))
    _, segs_a, toks_a = eng.transcribe_segments(x, max_new_tokens=nmax, language_token=sp["lang0"], prev_text=True)
    _, segs_b, toks_b = eng.transcribe_segments(x, max_new_tokens=nmax, language_token=sp["lang0"], prev_text=False)
    if toks_a == toks_b:
        print(f"seed {seed}: conditioning changes nothing ({len(toks_a)} tokens, {len(segs_a)} segments)", flush=True)
        continue
    rsegs, rkept, wins = WO.transcribe_timestamps(W, hp, lambda seek: oracle_lib.oracle_logmel(x, F, seek), x.size,
                                                  [sp["sot"], sp["lang0"], sp["transcribe"]], WO.RULES_WCPP, eng.token_text, n_max=nmax,
                                                  suppress=sup, suppress_first=[220, sp["eot"]], max_windows=16)
    ok = toks_a == [t for t in rkept if t != sp["eot"]]
    cond = [len(w["prompt"]) for w in wins]
    print(f"seed {seed}: differs with conditioning; oracle windows {len(wins)}, prompt lengths {cond}, min margin "
          f"{min(min(w['margins']) for w in wins):.2e}, product == oracle: {ok}", flush=True)
