#!/usr/bin/env python3
"""Pinning kit, step 1 of 3: write the INPUTS that someone who can run the reference feeds to it.

    python tools/make_ref_inputs.py DIR
    cargo run --release --example dump_vectors -- DIR [ggml-model.bin]      # bindings/rust/crispy-hip-sys, needs cargo + crates
    CRISPY_REF_VECTORS=DIR python -m pytest tests/test_reference_vectors.py [-m gpu]

Nothing here (or in this repository) can run the reference: no Rust toolchain, crates not vendored.
This script only lays down seeded inputs as raw little-endian f32 files -- trivially readable from Rust -- plus the
rnnoise-nu text models `nnnoiseless::RnnModel::from_read` parses, and a manifest.  The outputs the Rust example
writes beside them (`ref_*.f32` / `ref_*.json`) are what tests/test_reference_vectors.py compares the CPU oracle and
the HIP path with; with them in hand "parity unpinned" ends."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crispy_amd import rnn_weights, synth_audio  # noqa: E402


def main(out_dir: str) -> None:
    os.makedirs(out_dir, exist_ok=True)
    man = {"format": "raw little-endian float32", "rnnoise": [], "resampler": {}, "asr": {}}
    G = np.load(os.path.join(ROOT, "tests", "golden", "rnnoise_golden.npz"))
    for seed in (0, 1, 2):
        rnn_weights.save_rnnoise_nu_text(os.path.join(out_dir, f"rnnoise_model_seed{seed}.txt"), rnn_weights.synthetic_weights(seed))
    for kind in ("heavy_tail", "row_saturating"):          # trained-like and clamp-hitting weights (parity hardening, round 2)
        rnn_weights.save_rnnoise_nu_text(os.path.join(out_dir, f"rnnoise_model_{kind}.txt"), rnn_weights.extreme_weights(kind))

    def rn_case(name, model, x):
        x = np.ascontiguousarray(x, dtype="<f4").reshape(-1, 480)
        x.tofile(os.path.join(out_dir, f"rn_{name}_in.f32"))
        man["rnnoise"].append({"name": name, "model": model, "frames": int(x.shape[0]), "in": f"rn_{name}_in.f32",
                               "ref_out": f"ref_rn_{name}_out.f32", "ref_vad": f"ref_rn_{name}_vad.f32"})

    # BASELINE cfg 1: one 30 s clip, int16-range samples as RnnNoiseProcessor::push_sample hands them over (audio.rs:264)
    rn_case("cfg1", "rnnoise_model_seed0.txt", synth_audio.cfg1_clip(3000) * np.float32(32768.0))
    for seed in (0, 1, 2):                                   # the committed golden inputs (tests/golden/rnnoise_golden.npz)
        rn_case(f"golden_seed{seed}", f"rnnoise_model_seed{seed}.txt", G[f"seed{seed}/x"])
    for k in ("silence", "tone440", "whisper_quiet"):
        rn_case(f"golden_{k}", "rnnoise_model_seed0.txt", G[f"{k}/x"])
    x = synth_audio.stream_np(100, 300, silent=False) * np.float32(32768.0)
    rn_case("heavy_tail", "rnnoise_model_heavy_tail.txt", x)
    rn_case("row_saturating", "rnnoise_model_row_saturating.txt", x)
    # the same cfg 1 clip through the model BUILT INTO nnnoiseless (DenoiseState::new(), audio.rs:229): comparable once a
    # maintainer also exports that model as text and passes it with `--rnnoise-model`
    man["rnnoise_builtin"] = {"in": "rn_cfg1_in.f32", "frames": 3000, "ref_out": "ref_rn_cfg1_builtin_out.f32",
                              "ref_vad": "ref_rn_cfg1_builtin_vad.f32"}
    # rubato FftFixedIn(48000, 16000, 1024, 1, 1) in 1024-sample calls (commands/transcription.rs:198-208, 314-357)
    y = synth_audio.stream_np(200, 300, silent=False)[:48000 * 3 - 123]          # 3 s, not a multiple of 1024
    np.ascontiguousarray(y, dtype="<f4").tofile(os.path.join(out_dir, "rs_in48k.f32"))
    man["resampler"] = {"in": "rs_in48k.f32", "samples": int(y.size), "ref_out": "ref_rs_out16k.f32"}
    # whisper.cpp through whisper-rs on a supplied GGML file: greedy, TranscribeOptions::default()-like parameters
    z = synth_audio.clip16k_np(0, 464000)
    np.ascontiguousarray(z, dtype="<f4").tofile(os.path.join(out_dir, "asr_in16k.f32"))
    man["asr"] = {"in": "asr_in16k.f32", "samples": int(z.size), "ref": "ref_asr.json",
                  "note": "needs a real ggml model file: pass it to dump_vectors and to the test (CRISPY_REF_GGML)"}
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
    print(f"wrote {len(man['rnnoise'])} RNNoise cases, 1 resampler case, 1 ASR case into {out_dir}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ref_vectors")
