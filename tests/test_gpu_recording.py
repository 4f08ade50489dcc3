"""`crispy_asr_transcribe_recording` -- `run_transcription`'s chunk loop (src-tauri/src/commands/transcription.rs:249-302,
363-400, 468) with the chunks decoded side by side: the text must be byte for byte what the
reference-shaped loop gives (one engine call per 30 s chunk, trimmed texts joined with one space), the cancel flag must end
the call within one group (:251, :359, :402), and the progress hook must see every group (:285-299)."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(tmp_path_factory):
    from crispy_amd.asr import WhisperEngine
    from crispy_amd.ggml_io import synthetic_vocab, write_ggml
    from crispy_amd.mel_filters import whisper_mel_filters
    from crispy_amd.whisper_weights import HParams, synthetic_whisper_weights
    hp = HParams.tiny()
    W = synthetic_whisper_weights(hp, 0, sensitive=True)          # the tokens depend on the audio: every chunk says something else
    vocab = synthetic_vocab(hp.n_vocab)
    vocab[1000] = b"  "                                            # a token that is only white space (trimmed away at a chunk's ends)
    path = str(tmp_path_factory.mktemp("ggml_rec") / "tiny-sensitive.bin")
    write_ggml(path, hp, W, whisper_mel_filters(80), vocab, f16=False)
    eng = WhisperEngine(path)
    eng.set_precision(1)
    yield eng
    eng.close()


def _recording(n_chunks, tail):
    from crispy_amd import synth_audio
    parts = [synth_audio.clip16k_np(700 + i, 480000) for i in range(n_chunks)]
    parts[3][:] = 0.0                                              # a silent chunk in the middle
    if tail:
        parts.append(synth_audio.clip16k_np(900, tail))
    return np.concatenate(parts)


@pytest.mark.parametrize("timestamps", [False, True])
def test_recording_text_equals_the_chunk_by_chunk_loop(engine, timestamps):
    """20 chunks + a 7.3 s tail.  timestamps = False: plain greedy chunks; True: `TranscribeOptions::default()` -- whisper_full's
    seek loop per chunk (fallback off so that the run time is the decode, not the ladder of random-init logits)."""
    from crispy_amd.asr import transcribe_recording, transcribe_recording_serial
    x = _recording(20, 116800)
    kw = dict(max_new_tokens=12, timestamps=timestamps, fallback=False)
    serial = transcribe_recording_serial(engine, x, **kw)
    seen = []
    full = transcribe_recording(engine, x, progress=lambda d, t: seen.append((d, t)), with_result=True, **kw)
    assert full[0] == serial and len(serial) > 100
    assert seen == [(x.size, x.size)]                              # 21 chunks: one group
    # groups of 8 chunks: the same text, three progress reports, monotone, ending at the total
    seen.clear()
    assert transcribe_recording(engine, x, max_batch=8, progress=lambda d, t: seen.append((d, t)), **kw) == serial
    assert seen == [(8 * 480000, x.size), (16 * 480000, x.size), (x.size, x.size)]
    if timestamps:
        # segments carry recording time: chunk k's lie in [30 k, 30 (k + 1)] s; windows likewise in frames
        segs, wins = full[3], full[4]
        # (a random-init model puts timestamps anywhere in a chunk's 30 s, also behind the end of the 7.3 s tail chunk)
        assert segs and all(0.0 <= a <= b <= 21 * 30.0 + 0.02 for a, b, _ in segs)
        assert [s[0] for s in segs] == sorted(s[0] for s in segs)
        assert max(w["seek"] for w in wins) >= 20 * 3000
    # empty recording: empty text, no call-back (commands/transcription.rs:190-194)
    seen.clear()
    assert transcribe_recording(engine, np.zeros(0, np.float32), progress=lambda d, t: seen.append(d)) == "" and not seen


def test_recording_cancel_and_bad_arguments(engine):
    from crispy_amd import _native as N
    from crispy_amd.asr import Cancelled, make_opts, transcribe_recording
    x = _recording(6, 0)
    kw = dict(max_new_tokens=6, fallback=False)
    flag = C.c_int(1)
    with pytest.raises(Cancelled):                                 # set before the call: nothing runs
        transcribe_recording(engine, x, cancel=flag, **kw)
    flag = C.c_int(0)
    seen = []

    def on_progress(done, total):                                  # the host cancels after the first group
        seen.append(done)
        flag.value = 1

    with pytest.raises(Cancelled):
        transcribe_recording(engine, x, max_batch=2, cancel=flag, progress=on_progress, **kw)
    assert seen == [2 * 480000]                                    # "within one group": the second group never started
    flag.value = 0                                                 # the handle is usable afterwards
    assert transcribe_recording(engine, x, max_batch=2, cancel=flag, **kw) == transcribe_recording(engine, x, **kw)
    # cancelled between the windows of the seek loop: a watcher thread sets the flag while the one group is decoding
    import threading
    flag.value = 0
    th = threading.Timer(0.02, lambda: setattr(flag, "value", 1))
    th.start()
    try:
        with pytest.raises(Cancelled):
            for _ in range(200):                                   # (each call checks the flag; one of them sees it mid-flight or at its start)
                transcribe_recording(engine, x, cancel=flag, timestamps=True, max_new_tokens=40, fallback=False)
    finally:
        th.cancel()
    res = C.c_void_p()
    opts = make_opts(0, False, 4, False, True, carry_context=True)
    rc = engine._L.crispy_asr_transcribe_recording(engine._h, x.ctypes.data, x.size, C.byref(opts), 0, None, N.PROGRESS_FN(), None, C.byref(res))
    assert rc == -1 and b"carry_context" in engine._L.crispy_last_error()
    rc = engine._L.crispy_asr_transcribe_recording(engine._h, x.ctypes.data, x.size, None, -3, None, N.PROGRESS_FN(), None, C.byref(res))
    assert rc == -1
    rc = engine._L.crispy_asr_transcribe_recording(engine._h, None, 5, None, 0, None, N.PROGRESS_FN(), None, C.byref(res))
    assert rc == -1


def test_recording_in_one_call_against_one_call_per_chunk_timed(engine):
    """What the batch buys on the reference's own call shape -- `TranscribeOptions::default()`, i.e. whisper_full's seek loop
    with its full token budget per window (fallback off: on random-init logits every window would walk the temperature ladder,
    and the comparison would time that) -- for the 20.24 chunks of a ten-minute recording held in pageable host memory.
    Text equal, and the single call several times faster; the measured ratio is printed (>= 10 x)."""
    from crispy_amd.asr import transcribe_recording, transcribe_recording_serial
    x = _recording(20, 116800)
    kw = dict(timestamps=True, fallback=False)
    transcribe_recording(engine, x[:960000], **kw)                 # workspaces and captured steps of both shapes exist before anything is timed
    transcribe_recording_serial(engine, x[:480000], **kw)
    t0 = time.perf_counter()
    serial = transcribe_recording_serial(engine, x, **kw)
    t1 = time.perf_counter()
    once = transcribe_recording(engine, x, **kw)
    t2 = time.perf_counter()
    assert once == serial and len(serial) > 500
    ratio = (t1 - t0) / (t2 - t1)
    print(f"{x.size / 16000:.0f} s of audio, whisper_full per chunk: chunk by chunk {1e3 * (t1 - t0):.1f} ms, one call {1e3 * (t2 - t1):.1f} ms = {ratio:.1f} x")
    assert ratio >= 6.0, ratio
