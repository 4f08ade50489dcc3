"""Developer tool (GPU box): where the time of whisper_full's temperature ladder goes -- the scripted model of
tests/test_gpu_decision.py::test_temperature_ladder_on_a_scripted_model, each product call timed on its own, first and
second time (the second has every step graph captured).  MODE=0|1|2."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperEngine, transcribe_batch
from crispy_amd.ggml_io import synthetic_vocab, write_ggml
from crispy_amd.mel_filters import whisper_mel_filters
from crispy_amd.whisper_weights import HParams
from oracle import whisper_oracle as WO
from tests.scripted_model import script_rows, scripted_whisper_weights

hp = HParams.tiny()
sp = WO.special_tokens(hp.n_vocab)
BEG, EOT = sp["beg"], sp["eot"]
X, Y, REP = 1234, 2345, 777
beta = 1.0 - 1.0 * np.sqrt(2.0) / hp.n_text_state
rows = script_rows(2, [BEG, 1001, [(X, 1.0), (Y, beta)], 1003, BEG + 300, BEG + 300, EOT])
rows.update(script_rows(9, [BEG] + [REP] * 40 + [BEG + 100, BEG + 100, EOT]))
W = scripted_whisper_weights(hp, rows, gain=100.0)
path = os.path.join(tempfile.mkdtemp(), "ladder.bin")
write_ggml(path, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), f16=False)
eng = WhisperEngine(path)
eng.set_precision(int(os.environ.get("MODE", 1)))
x = synth_audio.clip16k_np(80, 16000 * 13)
for name, fn in (("single 13 s", lambda: eng.transcribe_segments(x, language_token=sp["lang0"])),
                 ("batch of 3", lambda: transcribe_batch(eng, [x[:16000 * 7], x, x[:16000 * 3]], language_token=sp["lang0"], timestamps=True, with_segments=True)),
                 ("batch of 64", lambda: transcribe_batch(eng, [x] * 64, language_token=sp["lang0"], timestamps=True, with_segments=True))):
    for rep in range(3):
        t0 = time.perf_counter()
        fn()
        print(f"MODE={os.environ.get('MODE', 1)} {name}: call {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
