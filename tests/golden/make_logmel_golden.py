"""Generates tests/golden/logmel_golden.npz with HuggingFace `WhisperFeatureExtractor`
(transformers, importable only in the build container): an independent implementation of the
front end whisper.cpp implements.  Inputs are regenerated from seeds by
crispy_amd.synth_audio.clip16k_np; every 7th frame of [80, 3000] is stored.

    python tests/golden/make_logmel_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transformers import WhisperFeatureExtractor  # noqa: E402

from crispy_amd import synth_audio  # noqa: E402

fe = WhisperFeatureExtractor()
blob = {"filters": np.ascontiguousarray(fe.mel_filters.T.astype(np.float32))}
# (seed, n_samples): 29 s (HF and whisper.cpp agree on every frame), 7.3 s (mostly padding), full 30 s
for seed, n in ((0, 464000), (1, 116800), (2, 480000)):
    x = synth_audio.clip16k_np(seed, n)
    ref = fe(x, sampling_rate=16000, return_tensors="np")["input_features"][0]
    blob[f"clip{seed}/n"] = np.int64(n)
    blob[f"clip{seed}/mel_every7"] = ref[:, ::7].astype(np.float32)
    blob[f"clip{seed}/mel_head"] = ref[:, :40].astype(np.float32)
    blob[f"clip{seed}/mel_tail"] = ref[:, -40:].astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "logmel_golden.npz"), **blob)
print("ok")
