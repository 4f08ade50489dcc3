"""The prompt of a decode call runs as multi-position steps (whisper_api.cpp: prefill / decoder_step with P > 1): P
positions of every clip in one pass through the decoder instead of P passes.  The claim is strong -- every row takes
exactly the arithmetic of the one-position step it replaces -- so the test is too: token ids AND the picked logits of
the following greedy decode are BIT-identical to the position-by-position prefill (CRISPY_ASR_PREFILL=seq), in the
precision modes, for one clip, a few, a full row range (64 x 4 = 256 rows), a chunked prompt (200 clips: 2 + 2
positions; a 37-token prompt of 20 clips: 25 + 12; the longest conditioned prompt, 228 tokens, of one clip in one step and
of three clips in 170 + 58) and per-clip language tokens."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from crispy_amd.asr import WhisperModel
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    m = WhisperModel(hp, synthetic_whisper_weights(hp, 0, sensitive=True))
    yield m
    m.close()


def _both(fn):
    old = os.environ.pop("CRISPY_ASR_PREFILL", None)
    try:
        a = fn()
        os.environ["CRISPY_ASR_PREFILL"] = "seq"
        b = fn()
    finally:
        os.environ.pop("CRISPY_ASR_PREFILL", None)
        if old is not None:
            os.environ["CRISPY_ASR_PREFILL"] = old
    return a, b


# (precision 2 -- the attentions normalise before they round, two more hand-offs per row -- on the shapes whose prompt steps
# put the rows of a clip through ONE cross-attention workgroup, attn_dec_x16g_kernel: more than 512 (row, head) pairs)
@pytest.mark.parametrize("precision,clips,n_prompt",
                         [(p, c, n) for p in (0, 1) for c, n in [(1, 4), (3, 4), (64, 4), (200, 4), (1, 37), (20, 37), (1, 228), (3, 228)]]
                         + [(2, 64, 4), (2, 20, 37), (2, 1, 228)])
def test_prompt_steps_are_bit_identical_to_the_position_by_position_prefill(model, precision, clips, n_prompt):
    import torch
    model.set_precision(precision)
    g = torch.Generator(device="cpu").manual_seed(100 + clips)
    enc = (torch.randn(clips, 1500, model.hp.n_audio_state, generator=g) * 0.7).to("cuda")
    torch.cuda.synchronize()
    rng = np.random.default_rng(n_prompt)
    prompt = [50258, 50259, 50359, 50363] if n_prompt == 4 else [50361] + rng.integers(0, 50000, n_prompt - 4).tolist() + [50258, 50259, 50359]
    assert len(prompt) == n_prompt
    (ta, na, la), (tb, nb, lb) = _both(lambda: model.decode_greedy_device(enc.data_ptr(), clips, prompt, 6))
    assert np.array_equal(ta, tb) and np.array_equal(na, nb)
    assert la.tobytes() == lb.tobytes(), f"picked logits differ: max {np.abs(la - lb).max():.3e}"
    assert len({tuple(r) for r in ta.tolist()}) > (1 if clips > 2 else 0)        # the clips do decode differently


@pytest.mark.parametrize("precision", [0, 1])
def test_per_clip_language_tokens_in_a_prompt_step(model, precision):
    import torch
    model.set_precision(precision)
    clips = 5
    g = torch.Generator(device="cpu").manual_seed(7)
    enc = (torch.randn(clips, 1500, model.hp.n_audio_state, generator=g) * 0.7).to("cuda")
    torch.cuda.synchronize()
    lang = [50259, 50260, 50261, 50259, 50300]
    (ta, na), (tb, nb) = _both(lambda: model.decode_greedy_lang_device(enc.data_ptr(), clips, [50258, 50259, 50359, 50363], lang, 6))
    assert np.array_equal(ta, tb) and np.array_equal(na, nb)
    # and the language token does reach the decoder: clip 0 and clip 3 share one, a solo run with another differs somewhere
    solo, _ = model.decode_greedy_lang_device(enc.data_ptr(), 1, [50258, 50259, 50359, 50363], [50300], 6)
    assert np.array_equal(ta[0], model.decode_greedy_lang_device(enc.data_ptr(), 1, [50258, 50259, 50359, 50363], [50259], 6)[0][0])
    assert solo.shape == (1, 6)
