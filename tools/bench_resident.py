#!/usr/bin/env python3
"""Catalog-size quantised models, resident vs inflated: seeded files of the shapes the reference's
catalog ships -- whisper-medium-q4_1 and ggml-large-v3-q5_0 (managers/model.rs:99,137) -- written on the box, then per
load flavour: load time, device bytes held by the model, and the product call for ONE 30 s chunk from host memory
(`crispy_asr_transcribe_tokens`: log-mel, encoder, cross K|V, prompt, 32 greedy tokens), split into its decode part.

    python tools/bench_resident.py medium:q4_1 large_v3:q5_0 [--tokens 32]

One JSON line per model on stdout."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    from crispy_amd import _native as N, synth_audio
    from crispy_amd.asr import WhisperEngine
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml_quantized
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, LazyWeights

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    new = int(sys.argv[sys.argv.index("--tokens") + 1]) if "--tokens" in sys.argv else 32
    x = synth_audio.clip16k_np(3, 480000)
    for spec in args or ["medium:q4_1"]:
        name, kind = spec.split(":")
        hp = getattr(HParams, name)()
        td = tempfile.mkdtemp(prefix="crispy_resident_", dir="/tmp")
        path = os.path.join(td, f"{name}-{kind}.bin")
        t0 = time.perf_counter()
        write_ggml_quantized(path, hp, LazyWeights(hp, 0, sensitive=True), whisper_mel_filters(hp.n_mels), synthetic_vocab(hp.n_vocab),
                             kind, keep=False)
        t_write = time.perf_counter() - t0
        sp = N.vocab_specials(hp.n_vocab)
        prompt = [sp.sot, sp.sot + 1, sp.transcribe, sp.notimestamps]
        out = {"model": f"{name} ({hp.n_audio_layer}+{hp.n_text_layer} layers, d {hp.n_text_state}) as {kind}",
               "file_mb": os.path.getsize(path) / 1e6, "write_s": t_write, "new_tokens": new, "flavours": {}}
        ids = {}
        for flavour in ("resident", "inflated_mode1", "inflated_mode0"):
            t0 = time.perf_counter()
            eng = WhisperEngine(path, resident=(flavour == "resident"))
            if flavour == "inflated_mode1":
                eng.set_precision(1)
            torch.cuda.synchronize()
            t_load = time.perf_counter() - t0
            mem = eng.memory_info()
            toks, _ = eng.transcribe_tokens([x], prompt, new)          # warm-up: workspaces, captured decode step
            ts, te = [], []
            for _ in range(3):
                t0 = time.perf_counter()
                toks, _ = eng.transcribe_tokens([x], prompt, new)
                ts.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                eng.encode([x])
                te.append(time.perf_counter() - t0)
            ids[flavour] = toks[0].tolist()
            full, enc = float(np.median(ts)), float(np.median(te))
            out["flavours"][flavour] = {"load_s": t_load, "weight_mb": mem["weight_bytes"] / 1e6,
                                        "quantised_mb": mem["quantised_bytes"] / 1e6, "scratch_mb": mem["scratch_bytes"] / 1e6,
                                        "one_chunk_ms": full * 1e3, "logmel_plus_encoder_ms": enc * 1e3,
                                        "decode_ms": (full - enc) * 1e3,
                                        "decode_ms_per_position": (full - enc) * 1e3 / (new + len(prompt))}
            eng.close()
            del eng
            torch.cuda.empty_cache()
        out["resident_ids_equal_inflated_mode1"] = ids["resident"] == ids["inflated_mode1"]
        r, i1 = out["flavours"]["resident"], out["flavours"]["inflated_mode1"]
        out["resident_over_file_bytes"] = r["weight_mb"] / out["file_mb"]
        out["decode_speedup_vs_inflated_mode0"] = out["flavours"]["inflated_mode0"]["decode_ms"] / r["decode_ms"]
        out["decode_speedup_vs_inflated_mode1"] = i1["decode_ms"] / r["decode_ms"]
        print(json.dumps(out), flush=True)
        os.remove(path)
        os.rmdir(td)


if __name__ == "__main__":
    main()
