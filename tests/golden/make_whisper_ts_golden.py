"""Generates tests/golden/whisper_tiny_ts_golden.npz: greedy decoding under the timestamp rules with HuggingFace
transformers -- `WhisperForConditionalGeneration` for the logits and `WhisperTimeStampLogitsProcessor`
(openai-whisper's ApplyTimestampRules) for the masking -- on the seeded synthetic Whisper-tiny weights.
This pins the RULES_OPENAI flavour of oracle/whisper_oracle.py:timestamp_rules; the whisper.cpp flavour differs
in three documented places and cannot be pinned here.

    python tests/golden/make_whisper_ts_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402
from transformers.generation.logits_process import WhisperTimeStampLogitsProcessor  # noqa: E402

from crispy_amd import synth_audio  # noqa: E402
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights  # noqa: E402
from hf_names import hf_name  # noqa: E402

hp = HParams.tiny()
W = synthetic_whisper_weights(hp, 0)
model = WhisperForConditionalGeneration(WhisperConfig()).eval()
sd = model.state_dict()
for n, v in W.items():
    sd[hf_name(n)].copy_(torch.from_numpy(v))
sd["proj_out.weight"].copy_(torch.from_numpy(W["decoder.token_embedding.weight"]))
model.load_state_dict(sd)

prompt = [50258, 50259, 50359]                      # sot, <|en|>, <|transcribe|>  (timestamps on)
suppress = list(range(50257 + 1, 50364))             # every special but EOT: sot, languages, tasks, notimestamps
suppress_first = [220, 50257]                        # suppress_blank: " " and EOT at the first position
cfg = SimpleNamespace(no_timestamps_token_id=50363, eos_token_id=50257, bos_token_id=50257,
                      max_initial_timestamp_index=50, _detect_timestamp_from_logprob=True)
proc = WhisperTimeStampLogitsProcessor(cfg, begin_index=len(prompt))
out = {}
for ci, (seed, n) in enumerate(((0, 464000), (5, 200000))):
    x = synth_audio.clip16k_np(seed, n)
    mel = WhisperFeatureExtractor()(x, sampling_rate=16000, return_tensors="pt")["input_features"]
    toks = list(prompt)
    picks, margins = [], []
    with torch.no_grad():
        enc = model.model.encoder(mel).last_hidden_state
        for step in range(40):
            lg = model(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([toks])).logits[0, -1].clone()
            lg[suppress] = -float("inf")
            if step == 0:
                lg[suppress_first] = -float("inf")
            sc = proc(torch.tensor([toks]), lg[None])[0].numpy()
            t = int(np.argmax(sc))
            top2 = np.partition(sc, -2)[-2:]
            picks.append(t); margins.append(float(top2[1] - top2[0]))
            toks.append(t)
            if t == 50257:
                break
    out[f"c{ci}_tokens"] = np.array(picks)
    out[f"c{ci}_margins"] = np.array(margins)
    out[f"c{ci}_clip"] = np.array([seed, n])
    print(ci, picks, np.round(margins, 3))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "whisper_tiny_ts_golden.npz"), prompt=np.array(prompt),
                    suppress=np.array(suppress), suppress_first=np.array(suppress_first), **out)
