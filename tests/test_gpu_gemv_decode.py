"""The decode step of the catalog widths (768 / 1024 / 1280: small, medium, large-v3 -- src-tauri/src/managers/model.rs:74-148)
at the reference's call shape, one chunk at a time (managers/transcription.rs:183-185), and at every other batch size:
`whisper_dec_gemv.hip` -- the six projections of a layer as matrix-vector products over dense f16 rows or resident ggml blocks,
LayerNorm computed in the consumer, 8 launches per layer instead of 11.

 * against the oracle of precision mode 1 (oracle/whisper_oracle.py DecoderCache(f16=True): ggml's mul_mat arithmetic) at the
   mode's bar, and against the same step through the skinny kernels (developer build, CRISPY_ASR_GEMV=0) -- two implementations
   of one arithmetic, not bit-identical, both at the bar;
 * a resident quantised engine == the same file inflated at load, ids and picked-logit BYTES (same instructions behind the
   weight fetch), for every ggml type the catalog uses;
 * a row's bits do not depend on the rows it shares a step with: 1 row == row 2 of 3, and == the row in batches of 6, 37 and
   130 (wide steps take the rows four at a time through the same kernels)."""
import numpy as np
import pytest

from tests.native_variant import library_variant

pytestmark = pytest.mark.gpu


def _hp(d, layers=2):
    from crispy_amd.whisper_weights import HParams
    return HParams(n_audio_state=d, n_audio_head=d // 64, n_audio_layer=1, n_text_state=d, n_text_head=d // 64, n_text_layer=layers)


@pytest.mark.parametrize("d,mode", [(768, 1), (1024, 1), (1280, 1), (768, 2)])
def test_gemv_step_against_the_oracle_and_the_skinny_kernels(d, mode):
    """mode 2 (ggml's rounding points inside the attentions) keeps the cross block as projection + one workgroup per head:
    its normalised probabilities need the row's maximum and sum before the first one is rounded."""
    import torch
    from crispy_amd.asr import WhisperModel
    from crispy_amd.whisper_weights import synthetic_whisper_weights
    from oracle import whisper_oracle as WO
    hp = _hp(d)
    W = synthetic_whisper_weights(hp, 5)
    rng = np.random.default_rng(d)
    B, n_new = 3, 8
    enc = (rng.standard_normal((B, 1500, d)) * 0.8).astype(np.float32)
    prompt = WO.default_prompt(hp.n_vocab, no_timestamps=True)
    d_enc = torch.from_numpy(enc).to("cuda:0")
    torch.cuda.synchronize()
    m = WhisperModel(hp, W)
    try:
        m.set_precision(mode)
        tg, _, lg = m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        tg2, _, lg2 = m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        solo, _, lsolo = m.decode_greedy_device(d_enc[2:3].contiguous().data_ptr(), 1, prompt, n_new)
    finally:
        m.close()
    assert np.array_equal(tg, tg2) and lg.tobytes() == lg2.tobytes()                  # deterministic
    assert np.array_equal(solo[0], tg[2]) and lsolo[0].tobytes() == lg[2].tobytes()    # alone = in the step, bit for bit
    with library_variant("dev", {"CRISPY_ASR_GEMV": "0"}):
        ms = WhisperModel(hp, W)
        try:
            ms.set_precision(mode)
            ts, _, ls = ms.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        finally:
            ms.close()
    best = np.zeros((B, n_new)); margin = np.zeros((B, n_new)); ids = np.zeros((B, n_new), np.int64)
    for b in range(B):
        dc = WO.DecoderCache(W, hp, enc[b], f16=True, attn16=(mode == 2))
        for t in prompt[:-1]:
            dc.step(t)
        tok = prompt[-1]
        for i in range(n_new):
            l = dc.step(tok)
            tok = int(tg[b][i])
            best[b, i] = l[tok]
            top = np.partition(l, -2)[-2:]
            margin[b, i] = top[1] - top[0]
            ids[b, i] = int(np.argmax(l))
    scale = np.abs(best).max()
    eg = (lg - best) / scale
    same = tg == ts
    es = (ls - best)[same] / scale
    print(f"d {d} mode {mode}: gemv rms {np.sqrt(np.mean(eg ** 2)):.2e} worst {np.abs(eg).max():.2e}; skinny rms {np.sqrt(np.mean(es ** 2)):.2e} "
          f"worst {np.abs(es).max():.2e}; forms agree on {int(same.sum())} of {same.size} picks")
    assert np.sqrt(np.mean(eg ** 2)) < 1.6e-4 and np.abs(eg).max() < 5e-4
    assert np.sqrt(np.mean(es ** 2)) < 1.6e-4 and np.abs(es).max() < 5e-4
    resolved = margin > 1e-3 * scale
    assert resolved.sum() >= B * n_new // 2
    assert np.array_equal(tg[resolved], ids[resolved]) and np.array_equal(ts[resolved], ids[resolved])


@pytest.mark.parametrize("d,kind", [(1024, "q4_1"), (1280, "q5_0"), (768, "q8_0"), (768, "q4_0"), (1024, "q5_1")])
def test_resident_blocks_equal_the_inflated_file_through_the_gemv_step(tmp_path, d, kind):
    """medium ships as q4_1, large-v3 as q5_0 (managers/model.rs:99,137); the other ggml types for completeness."""
    import torch
    from crispy_amd.asr import WhisperEngine
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml_quantized
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import synthetic_whisper_weights
    hp = _hp(d)
    W = synthetic_whisper_weights(hp, 7, sensitive=True)
    path = str(tmp_path / f"w{d}-{kind}.bin")
    write_ggml_quantized(path, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), kind)
    rng = np.random.default_rng(d + len(kind))
    B, n_new = 2, 10
    d_enc = torch.from_numpy((rng.standard_normal((B, 1500, d)) * 0.8).astype(np.float32)).to("cuda:0")
    torch.cuda.synchronize()
    prompt = [50258, 50259, 50359, 50363]
    res = WhisperEngine(path, resident=True)
    inf = WhisperEngine(path)
    try:
        inf.set_precision(1)
        tr, _, lr = res.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        ti, _, li = inf.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
        t1, _, l1 = res.decode_greedy_device(d_enc[1:2].contiguous().data_ptr(), 1, prompt, n_new)
    finally:
        res.close(); inf.close()
    assert np.array_equal(tr, ti) and lr.tobytes() == li.tobytes(), (tr, ti)
    assert np.array_equal(t1[0], tr[1]) and l1[0].tobytes() == lr[1].tobytes()
    assert len({tuple(t) for t in tr.tolist()}) == B


@pytest.mark.parametrize("d,kind", [(768, None), (1024, "q4_1"), (1280, "q5_0")])
def test_a_row_of_a_catalog_width_decodes_to_the_same_bits_in_any_batch(tmp_path, d, kind):
    """One clip, one answer for the models the app loads: a generated token's step of a catalog width runs the matrix-vector
    kernels at EVERY row count (gridDim.y takes the rows four at a time), not the skinny MFMA tiles above four rows -- so a
    clip's ids and picked-logit bytes are the same alone, as row 5 of 6, and in batches of 37 and 130 (the reference decodes
    one chunk per call, managers/transcription.rs:183-185; `crispy_asr_transcribe_recording` decodes up to 128 at once and
    has to say the same).  Dense f16 (small's width), resident q4_1 blocks (medium's width and type) and resident q5_0 blocks (large-v3's)."""
    import torch
    from crispy_amd.asr import WhisperEngine, WhisperModel
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml_quantized
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import synthetic_whisper_weights
    hp = _hp(d)
    W = synthetic_whisper_weights(hp, 11, sensitive=True)
    if kind:
        path = str(tmp_path / f"w{d}-{kind}.bin")
        write_ggml_quantized(path, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), kind)
        m = WhisperEngine(path, resident=True)
    else:
        m = WhisperModel(hp, W)
        m.set_precision(1)
    rng = np.random.default_rng(d)
    n_new = 6
    base = (rng.standard_normal((6, 1500, d)) * 0.8).astype(np.float32)
    prompt = [50258, 50259, 50359, 50363]
    try:
        ref = {}
        for B in (1, 6, 37, 130):
            enc = np.ascontiguousarray(base[np.arange(B) % 6])
            d_enc = torch.from_numpy(enc).to("cuda:0")
            torch.cuda.synchronize()
            t, _, l = m.decode_greedy_device(d_enc.data_ptr(), B, prompt, n_new)
            for r in range(B):
                key = r % 6
                if key not in ref:
                    ref[key] = (t[r].copy(), l[r].tobytes())
                assert np.array_equal(t[r], ref[key][0]) and l[r].tobytes() == ref[key][1], (B, r)
            del d_enc
        assert len({tuple(v[0].tolist()) for v in ref.values()}) >= 3      # the six clips do not all decode alike
    finally:
        m.close()


@pytest.mark.parametrize("opts", [dict(), dict(beam_size=3)], ids=["ladder_best_of_5", "beam_3"])
def test_fallback_and_beam_passes_of_a_catalog_width_do_not_depend_on_the_batch(tmp_path, opts):
    """whisper_full's fallback passes (five sampling decoders per clip over one cross K | V) and its beam search through the
    matrix-vector step: rows of a clip share its keys (`XattnArgs::group`), a pass of three clips is a step of 15 rows -- the
    wide form (LayerNorms and cross q as launches of their own, rows four at a time) -- a pass of one clip a step of 5.  On
    random-init weights every window walks the ladder.  Each clip of a 3-clip call == the clip alone: text, tokens, segments,
    window decisions."""
    from crispy_amd import synth_audio
    from crispy_amd.asr import WhisperEngine, transcribe_batch
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import synthetic_whisper_weights
    hp = _hp(768)
    W = synthetic_whisper_weights(hp, 21, sensitive=True)
    path = str(tmp_path / "w768.bin")
    write_ggml(path, hp, W, whisper_mel_filters(80), synthetic_vocab(hp.n_vocab), f16=True)
    eng = WhisperEngine(path)
    try:
        eng.set_precision(1)
        clips = [synth_audio.clip16k_np(500 + i, 16000 * (4 + 2 * i)) for i in range(3)]
        kw = dict(timestamps=True, with_segments=True, max_new_tokens=24, **opts)
        got = transcribe_batch(eng, clips, **kw)
        assert any(w["temperature"] > 0 for r in got for w in r[4]) or opts        # the ladder ran (or this is the beam case)
        for c in range(3):
            assert got[c] == transcribe_batch(eng, [clips[c]], **kw)[0], c
    finally:
        eng.close()
