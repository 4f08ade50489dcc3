"""The fused decode step (crispy_amd/csrc/whisper_dec_fused.hip): a generated token's decoder layer in three launches --
self-attention block, cross-attention block, MLP block, each with the preceding projection's all-to-all turned into
partial rows the next launch adds up.  It is the ONE form the product decodes a generated token of a dense Whisper-tiny /
-base model with in precision modes 1 / 2, whatever the batch (1 .. 512 rows): asserted here as "a row decodes to the same
ids and the same picked-logit bytes alone and in batches of 70, 129 and 512 rows" (the reference has
one engine and one answer per chunk: src-tauri/src/managers/transcription.rs:27,178).

Second implementation of the same arithmetic: the step as one launch per stage (`CRISPY_ASR_DECODE=stages`, a developer
knob that only libcrispy_hip_dev.so reads -- tests/native_variant.py).  The two forms add a row's partial sums in different
orders, so they are NOT bit-identical to each other; each is deterministic, and both sit at the mode's bar from the oracle
(oracle/whisper_oracle.py DecoderCache(f16=True): f16 LayerNorm outputs, f16 K | V caches, f16 operands of every product,
f32 accumulation -- ggml's arithmetic [UPSTREAM-RECALL]).
Reference call shape: engine.transcribe per token, src-tauri/src/managers/transcription.rs:183-185."""
import numpy as np
import pytest

from tests.native_variant import staged_decoder

pytestmark = pytest.mark.gpu


def _staged(hp, W, mode, d_enc_ptr, rows, prompt, n_new):
    """The same decode call through the developer build's one-launch-per-stage step."""
    from crispy_amd.asr import WhisperModel
    with staged_decoder():
        m = WhisperModel(hp, W)
        try:
            m.set_precision(mode)
            return m.decode_greedy_device(d_enc_ptr, rows, prompt, n_new)
        finally:
            m.close()


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
        tf, _, lf = m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        tf2, _, lf2 = m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        solo, _, lsolo = m.decode_greedy_device(d_enc[2:3].contiguous().data_ptr(), 1, prompt, n_new)
    finally:
        m.set_precision(0)
    ts, _, ls = _staged(hp, W, mode, d_enc.data_ptr(), B, prompt, n_new)
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
        tf, nf, lf = m.decode_greedy_device(d_enc.data_ptr(), rows, prompt, n_new)
        r = rows - 1
        solo, _, lsolo = m.decode_greedy_device(d_enc[r:r + 1].contiguous().data_ptr(), 1, prompt, n_new)
    finally:
        m.set_precision(0)
    ts, ns, ls = _staged(hp, W, 1, d_enc.data_ptr(), rows, prompt, n_new)
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


@pytest.mark.parametrize("mode", [1, 2])
def test_a_row_decodes_to_the_same_bits_in_batches_of_1_70_129_and_512(model, mode):
    """The product's own path selection, no override: the first 1 / 70 / 129 / 512 of 512 clips decode as ONE step batch each
    (the kernels pick other rows-per-workgroup shapes, grids of 6 to 8192 workgroups, plain and non-temporal K | V loads);
    every clip of a smaller batch must come out of every larger one with the same ids and the same picked-logit BYTES.
    512 rows is the step size BASELINE cfg 4 decodes with, 129 the first size that took the staged kernels in round 5."""
    import torch
    m, hp, W = model
    n_new = 12
    g = torch.Generator(device="cuda:0").manual_seed(77)
    d_enc = torch.randn(512, 1500, hp.n_audio_state, generator=g, device="cuda:0") * 0.8
    prompt = [50258, 50259, 50359, 50363]
    torch.cuda.synchronize()
    out = {}
    try:
        m.set_precision(mode)
        for rows in (1, 70, 129, 512):
            t, _, l = m.decode_greedy_device(d_enc.data_ptr(), rows, prompt, n_new)
            out[rows] = (t.copy(), l.copy())
        # and from the middle of the big batch: clip 300 alone
        t300, _, l300 = m.decode_greedy_device(d_enc[300:301].contiguous().data_ptr(), 1, prompt, n_new)
    finally:
        m.set_precision(0)
    t512, l512 = out[512]
    for rows in (1, 70, 129):
        t, l = out[rows]
        assert np.array_equal(t, t512[:rows]), (rows, np.nonzero((t != t512[:rows]).any(axis=1))[0][:8])
        assert l.tobytes() == l512[:rows].tobytes(), rows
    assert np.array_equal(t300[0], t512[300]) and l300[0].tobytes() == l512[300].tobytes()
    assert len({tuple(r) for r in t512.tolist()}) > 32           # the rows are different sequences, not one repeated (67 on random-init weights)


def test_a_batch_call_of_more_than_512_clips_takes_them_in_turns_and_says_the_same():
    """`crispy_asr_transcribe_batch` decodes at most 512 clips in lock step -- the widest step of the forms whose arithmetic per
    row is the row's alone -- and takes a larger batch in turns: clip 0, 511, 512 and 519 of a 520-clip call come out as from a
    call of their own (tokens and language; a 600-chunk recording says what 600 single calls say)."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperModel, transcribe_batch
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    m = WhisperModel(hp, synthetic_whisper_weights(hp, 4, sensitive=True))
    try:
        m.set_precision(1)
        clips = [synth_audio.clip16k_np(300 + (i % 7), 16000 * 2 + 160 * (i % 5)) for i in range(520)]
        kw = dict(max_new_tokens=6, timestamps=False)
        got = transcribe_batch(m, clips, **kw)
        assert len(got) == 520
        for c in (0, 511, 512, 519):
            assert got[c] == transcribe_batch(m, [clips[c]], **kw)[0], c
        assert len({tuple(g[1]) for g in got}) >= 3
    finally:
        m.close()
