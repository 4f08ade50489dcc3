"""Developer tool (GPU box): the reference's literal call on ONE 28 s clip -- crispy_asr_transcribe, one greedy pass per
window (temperature_inc < 0), random-init tiny, mode 1 -- for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_single.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperModel, transcribe_batch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

hp = HParams.tiny()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(1)
x = synth_audio.clip16k_np(0, 16000 * 28)
for rep in range(4):
    t0 = time.perf_counter()
    (text, toks, lang, segs, wins), = transcribe_batch(m, [x], timestamps=True, with_segments=True, fallback=False)
    print(f"call {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {len(toks)} tokens, windows {[(w['seek'], w['n_tokens']) for w in wins]}", flush=True)
