"""The fused decode step (crispy_amd/csrc/whisper_dec_fused.hip): a generated token's decoder layer in three launches --
self-attention block, cross-attention block, MLP block, each with the preceding projection's all-to-all turned into
partial rows the next launch adds up -- against the same step as one launch per stage (CRISPY_ASR_DECODE=stages, a test
hook of the library) and against the oracle of the mode (oracle/whisper_oracle.py DecoderCache(f16=True): f16 LayerNorm
outputs, f16 K | V caches, f16 operands of every product, f32 accumulation -- ggml's arithmetic [UPSTREAM-RECALL]).
Reference call shape: engine.transcribe per token, src-tauri/src/managers/transcription.rs:183-185.

The two launch forms add a row's partial sums in different orders, so they are NOT bit-identical to each other; each is
deterministic and independent of the batch (asserted), and both sit at the mode's bar from the oracle (asserted)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _path(stages, fn):
    old = os.environ.pop("CRISPY_ASR_DECODE", None)
    try:
        if stages:
            os.environ["CRISPY_ASR_DECODE"] = "stages"
        return fn()
    finally:
        os.environ.pop("CRISPY_ASR_DECODE", None)
        if old is not None:
            os.environ["CRISPY_ASR_DECODE"] = old


@pytest.fixture(scope="module", params=["tiny", "base"])
def model(request):
    from crispy_amd.asr import WhisperModel
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = getattr(HParams, request.param)()
    W = synthetic_whisper_weights(hp, 5)       # plain fan-in-scaled weights: no amplification between the forms
    m = WhisperModel(hp, W)
    yield m, hp, W
    m.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_fused_step_against_the_staged_step_and_the_oracle(model, mode):
    """5 clips x 10 picks behind a 4-token prompt: picked logits of both launch forms against the oracle that follows the
    fused form's picks; ids equal between the forms and with the oracle wherever its top-2 margin exceeds the bar."""
    import torch
    from oracle import whisper_oracle as WO
    m, hp, W = model
    rng = np.random.default_rng(21)
    B, n_new = 5, 10
    enc = (rng.standard_normal((B, 1500, hp.n_audio_state)) * 0.8).astype(np.float32)
    prompt = WO.default_prompt(hp.n_vocab, no_timestamps=True)
    d_enc = torch.from_numpy(enc).to("cuda:0")
    torch.cuda.synchronize()
    try:
        m.set_precision(mode)
        tf, _, lf = _path(False, lambda: m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new))
        tf2, _, lf2 = _path(False, lambda: m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new))
        ts, _, ls = _path(True, lambda: m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new))
        solo, _, lsolo = _path(False, lambda: m.decode_greedy_device(d_enc[2:3].contiguous().data_ptr(), 1, prompt, n_new))
    finally:
        m.set_precision(0)
    assert np.array_equal(tf, tf2) and lf.tobytes() == lf2.tobytes()                 # deterministic
    assert np.array_equal(solo[0], tf[2]) and lsolo[0].tobytes() == lf[2].tobytes()   # alone = in the batch, bit for bit
    best = np.zeros((B, n_new)); margin = np.zeros((B, n_new)); ids = np.zeros((B, n_new), np.int64)
    for b in range(B):
        dc = WO.DecoderCache(W, hp, enc[b], f16=True, attn16=(mode == 2))
        for t in prompt[:-1]:
            dc.step(t)
        tok = prompt[-1]
        for i in range(n_new):
            l = dc.step(tok)
            tok = int(tf[b][i])
            best[b, i] = l[tok]
            top = np.partition(l, -2)[-2:]
            margin[b, i] = top[1] - top[0]
            ids[b, i] = int(np.argmax(l))
    scale = np.abs(best).max()
    ef = (lf - best) / scale
    rms_f = float(np.sqrt(np.mean(ef ** 2)))
    same = tf == ts
    es = (ls - best)[same] / scale                        # the staged form's picked logit where it picked the same token
    rms_s = float(np.sqrt(np.mean(es ** 2)))
    print(f"mode {mode} {hp.n_text_state}: fused rms {rms_f:.2e} worst {np.abs(ef).max():.2e}; staged rms {rms_s:.2e} worst {np.abs(es).max():.2e}; "
          f"forms agree on {int(same.sum())} of {same.size} picks")
    assert rms_f < 1.6e-4 and np.abs(ef).max() < 5e-4, (rms_f, np.abs(ef).max())
    assert rms_s < 1.6e-4 and np.abs(es).max() < 5e-4, (rms_s, np.abs(es).max())
    resolved = margin > 1e-3 * scale
    assert resolved.sum() >= B * n_new // 2, resolved.sum()
    assert np.array_equal(tf[resolved], ids[resolved])
    assert np.array_equal(ts[resolved], ids[resolved])


@pytest.mark.parametrize("n_new,rows", [(140, 3), (300, 2), (40, 70)])
def test_fused_step_over_every_key_class_and_many_rows(model, n_new, rows):
    """The self-attention of the fused step holds its keys in 1 / 2 / 4 register slots per wave (<= 128 / 256 / 512 positions,
    chosen per decode call): 144 and 304 positions run the two wider forms; 70 rows a grid of more workgroups than CUs in
    the MLP block.  Mode 1, ids against the staged form: the forms agree on long prefixes (at least 12 picks in nine rows of
    ten, 30 in the median row) with picked logits equal to the mode's bar, and a row equals its solo run bit for bit."""
    import torch
    m, hp, W = model
    rng = np.random.default_rng(n_new)
    enc = (rng.standard_normal((rows, 1500, hp.n_audio_state)) * 0.8).astype(np.float32)
    prompt = [50258, 50259, 50359, 50363]
    d_enc = torch.from_numpy(enc).to("cuda:0")
    torch.cuda.synchronize()
    try:
        m.set_precision(1)
        tf, nf, lf = _path(False, lambda: m.decode_greedy_device(d_enc.data_ptr(), rows, prompt, n_new))
        ts, ns, ls = _path(True, lambda: m.decode_greedy_device(d_enc.data_ptr(), rows, prompt, n_new))
        r = rows - 1
        solo, _, lsolo = _path(False, lambda: m.decode_greedy_device(d_enc[r:r + 1].contiguous().data_ptr(), 1, prompt, n_new))
    finally:
        m.set_precision(0)
    assert np.array_equal(solo[0], tf[r]) and lsolo[0].tobytes() == lf[r].tobytes()
    agree = []
    for b in range(rows):
        d = np.nonzero(tf[b] != ts[b])[0]
        k = int(d[0]) if d.size else n_new
        agree.append(k)
        # up to the first pick the forms disagree on, their picked logits are the same numbers to the mode's bar
        if k:
            scale = float(np.abs(ls[b][:k]).max())
            assert np.abs(lf[b][:k] - ls[b][:k]).max() <= 6e-4 * scale, (b, k)
    print(f"{hp.n_text_state} {n_new} new x {rows} rows: forms agree on the first {min(agree)} .. {max(agree)} picks")
    # (a row whose top two logits sit closer than the forms' difference in accumulation order parts ways there -- one of
    # 70 rows did at its 6th pick -- and everything behind that pick is another sequence)
    assert np.median(agree) >= min(n_new, 30) and np.mean(np.asarray(agree) >= 12) >= 0.9, agree
