"""Developer tool (GPU box): the 64-clip ladder of bench.py's asr.batch_ladder (random-init tiny, mode 1, every window walking the
whole temperature ladder) on its own, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_ladder.py`.  CLIPS=64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispy_amd import synth_audio
from crispy_amd.asr import WhisperModel, transcribe_batch
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights

hp = HParams.tiny()
m = WhisperModel(hp, synthetic_whisper_weights(hp, 0))
m.set_precision(1)
n = int(os.environ.get("CLIPS", 64))
clips = [synth_audio.clip16k_np(i, 16000 * 28) for i in range(n)]
for rep in range(3):
    t0 = time.perf_counter()
    res = transcribe_batch(m, clips, timestamps=True, with_segments=True)
    print(f"{n} clips: call {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms, windows {sum(len(r[4]) for r in res)}", flush=True)
