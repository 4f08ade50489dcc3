"""CPU tests of the C-ABI's robustness promises (include/crispy_hip.h "Conventions"): nothing throws across the
boundary (the reference host builds with panic=abort: /root/reference/src-tauri/Cargo.toml:10-20), model files are
validated before anything is sized from them, and the pure (device-free) entry points."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crispy_amd", "csrc")


def _lib():
    from crispy_amd import _native as N
    return N, N.lib()


# ---- exception guard ---------------------------------------------------------------------------------------------
def test_exception_guard_turns_exceptions_into_status_codes():
    N, L = _lib()
    assert L.crispy_selftest_exception_guard(0) == 0
    for kind, code, word in ((1, -4, b"bad_alloc"), (2, -4, b"length_error"), (3, -3, b"selftest"),
                             (4, -3, b"non-standard"), (5, -4, b"std::")):
        assert L.crispy_selftest_exception_guard(kind) == code, kind
        msg = L.crispy_last_error()
        assert b"crispy_selftest_exception_guard" in msg and word in msg, msg


def test_every_entry_point_definition_is_guarded():
    """Each extern "C" definition with a body of more than one statement is a function-try-block closed by
    CRISPY_CATCH_*; the few one-liners left out are listed here and cannot throw."""
    trivially_safe = {"crispy_last_error", "crispy_version", "crispy_abi_version", "crispy_rn_destroy", "crispy_rn_n_streams",
                      "crispy_rn_frames_per_launch"}
    hdr = open(os.path.join(ROOT, "include", "crispy_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(crispy_[a-z0-9_]+)\s*\(", hdr))
    guarded = set()
    for f in os.listdir(CSRC):
        if f.endswith((".cpp", ".hip")):
            src = open(os.path.join(CSRC, f)).read()
            for m in re.finditer(r'^(?:int|void|long) (crispy_\w+)\([^;{]*\)\s*try\s*\{', src, flags=re.M):
                name = m.group(1)
                assert re.search(r'CRISPY_CATCH_(RET|VOID)\("%s"\)' % name, src), f"{name}: try without its catch"
                guarded.add(name)
    assert declared - guarded <= trivially_safe, sorted(declared - guarded - trivially_safe)


# ---- rnnoise-nu model files, parsed in C++ (crispy_rn_weights_from_file) ----------------------------------------------
def _parse(L, path):
    blob = np.zeros(87503, np.int8)
    rc = L.crispy_rn_weights_from_file(str(path).encode(), blob.ctypes.data, blob.size)
    return rc, blob


def test_native_model_file_parser_matches_the_python_loader(tmp_path):
    """Same cases as test_weight_blob_layout_and_text_round_trip, through the C ABI a Rust host can call."""
    from crispy_amd import rnn_weights as RW
    N, L = _lib()
    w = RW.synthetic_weights(3)
    w[:5] = (-128, 127, 0, -1, 1)
    p = tmp_path / "model.txt"
    RW.save_rnnoise_nu_text(str(p), w)
    rc, blob = _parse(L, p)
    assert rc == 0, L.crispy_last_error()
    assert np.array_equal(blob, w) and np.array_equal(blob, RW.load_rnnoise_nu_text(str(p)))
    # CRLF line ends and irregular spacing are whitespace like any other
    q = tmp_path / "crlf.txt"
    head, body = open(p, "rb").read().split(b"\n", 1)
    q.write_bytes(head + b"\r\n" + body.replace(b"\n", b"\r\n").replace(b" ", b"  ", 50))
    rc, blob = _parse(L, q)
    assert rc == 0 and np.array_equal(blob, w)

    def bad(text, word):
        b = tmp_path / "bad.txt"
        b.write_text(text)
        rc, _ = _parse(L, b)
        assert rc == -5, (rc, word)
        assert word in L.crispy_last_error(), L.crispy_last_error()

    good = open(p).read()
    bad("not a model\n1 2 3\n", b"not an rnnoise-nu model file")
    bad(good[:2000], b"truncated")
    bad(good.replace("42 24 0", "42 25 0", 1), b"expected 42x24")
    bad(good.replace("42 24 0", "42 24 1", 1), b"unsupported activation")
    lines = good.split("\n")
    lines[2] = "300 " + lines[2].split(" ", 1)[1]
    bad("\n".join(lines), b"out of int8 range")
    lines[2] = "1.5 " + lines[2].split(" ", 1)[1]
    bad("\n".join(lines), b"truncated")
    assert L.crispy_rn_weights_from_file(b"/nonexistent/model.txt", blob.ctypes.data, blob.size) == -5
    assert L.crispy_rn_weights_from_file(str(p).encode(), blob.ctypes.data, 100) == -1
    assert L.crispy_rn_weights_from_file(None, blob.ctypes.data, blob.size) == -1


def test_create_from_file_reports_parse_errors_before_touching_a_device(tmp_path):
    N, L = _lib()
    h = C.c_void_p()
    b = tmp_path / "bad.txt"
    b.write_text("rnnoise-nu model file version 2\n")
    assert L.crispy_rn_create_from_file(str(b).encode(), 4, 0, C.byref(h)) == -5 and not h.value
    assert L.crispy_rn_create_from_file(str(b).encode(), 4, 0, None) == -1


# ---- whisper.cpp vocabulary specials (English-only models) -------------------------------------------------
def test_special_token_ids_for_en_multilingual_and_large_v3():
    """whisper.cpp `whisper_vocab`: defaults are the .en ids; multilingual shifts eot/sot by one and the tokens
    behind the language block by one more per extra language."""
    from oracle import whisper_oracle as WO
    N, L = _lib()
    want = {
        51864: dict(eot=50256, sot=50257, lang0=50258, n_lang=0, translate=50357, transcribe=50358, solm=50359,
                    prev=50360, nosp=50361, notimestamps=50362, beg=50363, multilingual=0),
        51865: dict(eot=50257, sot=50258, lang0=50259, n_lang=99, translate=50358, transcribe=50359, solm=50360,
                    prev=50361, nosp=50362, notimestamps=50363, beg=50364, multilingual=1),
        51866: dict(eot=50257, sot=50258, lang0=50259, n_lang=100, translate=50359, transcribe=50360, solm=50361,
                    prev=50362, nosp=50363, notimestamps=50364, beg=50365, multilingual=1),
    }
    for nv, w in want.items():
        sp = N.AsrSpecials()
        assert L.crispy_asr_vocab_specials(nv, C.byref(sp)) == 0
        got = {k: getattr(sp, k) for k, _ in sp._fields_}
        assert got == w, (nv, got)
        assert w["beg"] + 1501 == nv                       # <|0.00|> ... <|30.00|> are the last 1501 ids
        o = WO.special_tokens(nv)
        assert (o["eot"], o["sot"], o["n_lang"], o["translate"], o["transcribe"], o["solm"], o["prev"], o["nosp"],
                o["not_"], o["beg"]) == (w["eot"], w["sot"], w["n_lang"], w["translate"], w["transcribe"], w["solm"],
                                         w["prev"], w["nosp"], w["notimestamps"], w["beg"])
    assert WO.default_prompt(51864, no_timestamps=True) == [50257, 50362]          # .en: no language, no task
    assert WO.default_prompt(51865, no_timestamps=True) == [50258, 50259, 50359, 50363]
    assert WO.default_prompt(51866) == [50258, 50259, 50360]
    assert L.crispy_asr_vocab_specials(1000, C.byref(N.AsrSpecials())) == -1
    # language codes -> tokens (whisper.cpp's whisper_lang_id order); pure function, no device
    tok = C.c_int(-1)
    for nv, code, want in ((51865, b"en", 50259), (51865, b"zh", 50260), (51865, b"de", 50261), (51865, b"su", 50259 + 98),
                           (51866, b"yue", 50259 + 99), (51864, b"en", 0), (51865, b"auto", 0), (51865, b"", 0)):
        assert L.crispy_asr_language_token(nv, code, C.byref(tok)) == 0 and tok.value == want, (nv, code, tok.value)
    assert L.crispy_asr_language_token(51865, b"yue", C.byref(tok)) == -1        # 99 languages only
    assert L.crispy_asr_language_token(51865, b"xx", C.byref(tok)) == -1
    assert L.crispy_asr_language_token(51864, b"de", C.byref(tok)) == -6         # English-only vocabulary
    assert L.crispy_asr_language_token(51865, None, C.byref(tok)) == -1


# ---- GGML loader: nothing is sized from numbers a corrupt file supplies ---------------------------------------------
def _ggml_header(n_vocab=51865, d=384, n_mels=80, vocab_entries=0):
    hp = [n_vocab, 1500, d, d // 64, 4, 448, d, d // 64, 4, n_mels, 1]
    b = struct.pack("<I11i", 0x67676d6c, *hp)
    b += struct.pack("<2i", n_mels, 201) + np.zeros(n_mels * 201, np.float32).tobytes()
    b += struct.pack("<i", vocab_entries)
    return b


def _tensor(name, dims, ttype=0, data=b""):
    nb = name.encode()
    return struct.pack("<3i", len(dims), len(nb), ttype) + struct.pack(f"<{len(dims)}i", *dims) + nb + data


def test_ggml_loader_rejects_corrupt_files_without_allocating_from_them(tmp_path):
    N, L = _lib()
    h = C.c_void_p()

    def load(blob):
        p = tmp_path / "m.bin"
        p.write_bytes(blob)
        return L.crispy_asr_load(str(p).encode(), 0, C.byref(h))

    assert load(b"abcd") == -5 and b"bad magic" in L.crispy_last_error()
    assert load(_ggml_header(n_vocab=2 ** 31 - 1)) == -5 and b"n_vocab" in L.crispy_last_error()
    assert load(_ggml_header(vocab_entries=-3)) == -5
    if L.crispy_device_count() == 0:
        # everything below needs the engine object, i.e. a device: without one the load stops at NO_DEVICE, which is
        # itself the "no CPU fallback" contract
        assert load(_ggml_header() + _tensor("encoder.conv1.bias", [2 ** 30, 2 ** 30, 4])) == -2
        return
    # a shape whose element count overflows / is absurd, an unknown tensor, a known tensor with the wrong count
    assert load(_ggml_header() + _tensor("encoder.conv1.bias", [2 ** 30, 2 ** 30, 4])) == -5
    assert b"overflows" in L.crispy_last_error()
    assert load(_ggml_header() + _tensor("encoder.bogus", [4])) == -5 and b"unknown tensor" in L.crispy_last_error()
    assert load(_ggml_header() + _tensor("encoder.conv1.bias", [2 ** 30])) == -5
    assert b"imply 384" in L.crispy_last_error()
    assert not h.value


@pytest.mark.gpu
def test_ggml_loader_rejects_corrupt_files_on_the_gpu(tmp_path):
    test_ggml_loader_rejects_corrupt_files_without_allocating_from_them(tmp_path)
