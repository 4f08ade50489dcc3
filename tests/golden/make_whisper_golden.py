"""Generates tests/golden/whisper_tiny_golden.npz with HuggingFace transformers
(`WhisperForConditionalGeneration`, importable only in the build container) loaded with the seeded
synthetic Whisper-tiny weights of crispy_amd.whisper_weights (seed 0) -- an independent fp32
implementation of the graph the reference's whisper.cpp engine runs.

Stored: every 25th row of the encoder output for clip16k_np(0, 464000), the logits summary of the
4-token prompt, and 12 greedily decoded tokens (no suppression) with their logits and margins.

    python tests/golden/make_whisper_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402

from crispy_amd import synth_audio  # noqa: E402
from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hf_names import hf_name  # noqa: E402


hp = HParams.tiny()
W = synthetic_whisper_weights(hp, 0)
model = WhisperForConditionalGeneration(WhisperConfig()).eval()   # WhisperConfig() defaults are the tiny dims
sd = model.state_dict()
for n, v in W.items():
    sd[hf_name(n)].copy_(torch.from_numpy(v))
sd["proj_out.weight"].copy_(torch.from_numpy(W["decoder.token_embedding.weight"]))
model.load_state_dict(sd)

x = synth_audio.clip16k_np(0, 464000)
mel = WhisperFeatureExtractor()(x, sampling_rate=16000, return_tensors="pt")["input_features"]
prompt = [50258, 50259, 50359, 50363]
with torch.no_grad():
    enc = model.model.encoder(mel).last_hidden_state
    toks = list(prompt)
    picks, best, margin = [], [], []
    for _ in range(12):
        lg = model(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([toks])).logits[0, -1].numpy()
        t = int(np.argmax(lg))
        top2 = np.partition(lg, -2)[-2:]
        picks.append(t); best.append(float(lg[t])); margin.append(float(top2[1] - top2[0]))
        toks.append(t)
    lg_prompt = model(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([prompt])).logits[0].numpy()
enc = enc[0].numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "whisper_tiny_golden.npz"),
                    enc_rows=enc[::25].astype(np.float32), enc_mean_abs=np.float64(np.abs(enc).mean()),
                    enc_sum=np.float64(enc.astype(np.float64).sum()),
                    prompt=np.array(prompt), prompt_logits_sample=lg_prompt[:, ::997].astype(np.float32),
                    prompt_argmax=lg_prompt.argmax(-1), greedy_tokens=np.array(picks), greedy_logits=np.array(best),
                    greedy_margin=np.array(margin))
print("tokens", picks, "margins", np.round(margin, 4))
