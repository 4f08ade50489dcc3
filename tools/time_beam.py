"""Developer tool (GPU box): what a beam-search position costs next to a greedy one, through the product call
(`crispy_asr_transcribe_batch`, fallback off) on a scripted Whisper-tiny whose window is <|0.00|> + 160 text tokens with a
two- or three-way near tie at every fourth position (the beams stay apart and keep swapping cache rows) + a closing
timestamp pair.  The slope between a 40- and a 140-token limit is the cost of a position: the encoder, the prompt and
the call's fixed costs cancel.  1 and 16 clips; precision mode 1."""
import os, sys, tempfile, time
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
rng = np.random.default_rng(5)
n_text = 160
toks = rng.choice(np.arange(1000, 40000), size=3 * n_text, replace=False).tolist()
seq = [BEG]
for i in range(n_text):
    if i % 4 == 1:
        k = 2 + (i // 4) % 2
        w = [1.0] + [1.0 - float(rng.uniform(0.3, 2.5)) * np.sqrt(2.0) / hp.n_text_state for _ in range(k - 1)]
        seq.append([(toks[3 * i + j], w[j]) for j in range(k)])
    else:
        seq.append(toks[3 * i])
seq += [BEG + 600, BEG + 600, EOT]
W = scripted_whisper_weights(hp, script_rows(2, seq), gain=100.0)
path = os.path.join(tempfile.mkdtemp(), "beam.bin")
write_ggml(path, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), f16=False)
eng = WhisperEngine(path)
eng.set_precision(1)
x = synth_audio.clip16k_np(80, 16000 * 13)


def timed(clips, limit, **kw):
    transcribe_batch(eng, clips, timestamps=True, language_token=sp["lang0"], max_new_tokens=limit, fallback=False, **kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = transcribe_batch(eng, clips, timestamps=True, language_token=sp["lang0"], max_new_tokens=limit, fallback=False, with_segments=True, **kw)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, [r[4][0]["n_tokens"] for r in res]


for n_clips in (1, 16):
    clips = [x] * n_clips
    slope = {}
    for name, kw in (("greedy", {}), ("beam2", dict(beam_size=2)), ("beam5", dict(beam_size=5)), ("beam8", dict(beam_size=8))):
        t40, n40 = timed(clips, 40, **kw)
        t140, n140 = timed(clips, 140, **kw)
        slope[name] = (t140 - t40) / (n140[0] - n40[0])
        print(f"{n_clips:3d} clips {name:7s}: {t40:8.2f} ms at {n40[0]} tokens, {t140:8.2f} ms at {n140[0]} -> {slope[name] * 1e3:7.1f} us per position", flush=True)
    print(f"{n_clips:3d} clips: beam5 / greedy = {slope['beam5'] / slope['greedy']:.2f}, beam2 / greedy = {slope['beam2'] / slope['greedy']:.2f}, "
          f"beam8 / greedy = {slope['beam8'] / slope['greedy']:.2f}", flush=True)
