"""The committed oracle results the full-size GPU tests read (tests/golden/oracle_cache, tests/oracle_cache.py): every entry the
suite will ask for exists with the fingerprint of TODAY's inputs (a changed seed, script or oracle signature must regenerate
the cache, not silently fall back to minutes of numpy on the GPU box), and a cheap case recomputed here equals its entry."""
import json
import os

import numpy as np


def test_every_scripted_case_has_a_current_cache_entry_and_one_recomputes_to_it():
    from tests import oracle_cases as OC
    from tests.oracle_cache import CACHE_DIR, _dec, _enc, fingerprint
    cases = OC.scripted_cases()
    for W, hp, n, mode, kw in cases:
        fp = fingerprint("scripted_whisper_full", W["decoder.positional_embedding"], W["decoder.token_embedding.weight"], W["decoder.ln.weight"],
                         n, mode, kw)
        path = os.path.join(CACHE_DIR, f"scripted_whisper_full_{fp[:12]}.json")
        assert os.path.exists(path), (n, mode, kw, "run tests/golden/make_oracle_cache.py")
        assert json.load(open(path))["fingerprint"] == fp
    # the no-speech model's run is half a second of oracle: recomputed, it must be what the cache holds
    W, hp, n, mode, kw = cases[3]
    os.environ["CRISPY_ORACLE_CACHE"] = "off"
    try:
        fresh = OC.scripted_ref(W, hp, n, mode, **kw)
    finally:
        del os.environ["CRISPY_ORACLE_CACHE"]
    held = OC.scripted_ref(W, hp, n, mode, **kw)
    assert json.dumps(_enc(fresh), sort_keys=True) == json.dumps(_enc(held), sort_keys=True)
    assert held[0] == [] and len(held[2]) == 2 and all(w["is_no_speech"] for w in held[2])


def test_catalog_cache_entries_are_current():
    from tests import oracle_cases as OC
    from tests.oracle_cache import CACHE_DIR, fingerprint
    for name in ("small", "medium", "large_v3"):
        path = os.path.join(CACHE_DIR, f"catalog_{name}.json")
        assert os.path.exists(path), (name, "run tests/golden/make_oracle_cache.py catalog")
        blob = json.load(open(path))
        assert blob["fingerprint"] == fingerprint("catalog_case", name, OC.CATALOG_ROWS, 3, 77, 160000)
        v = blob["value"]
        assert v["ref_rows"]["shape"][0] == OC.CATALOG_ROWS and len(v["picks"]) == 3 and v["peak"] > 0


def test_cache_round_trip_keeps_what_tests_compare():
    from tests.oracle_cache import _dec, _enc
    v = {"a": np.float32(1.5), "b": [np.int64(3), (1, 2)], "c": b"\xff w1", "d": np.arange(6, dtype=np.float32).reshape(2, 3), "e": float("-inf"),
         "big": np.zeros(60000, np.float32), "obj": object()}
    r = _dec(json.loads(json.dumps(_enc(v))))
    assert r["a"] == 1.5 and r["b"] == [3, [1, 2]] and r["c"] == b"\xff w1" and np.array_equal(r["d"], v["d"]) and r["e"] == float("-inf")
    assert "__dropped__" in r["big"] and "__dropped__" in r["obj"]
