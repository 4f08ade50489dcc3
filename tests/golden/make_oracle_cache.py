"""Computes and commits the slow, GPU-independent oracle results of the full-size GPU tests (tests/oracle_cases.py) into
tests/golden/oracle_cache/*.json.  CPU only; about ten minutes (the three catalog encoders at full depth are most of it).

    python tests/golden/make_oracle_cache.py [scripted] [catalog]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CRISPY_ORACLE_CACHE"] = "write"

from tests import oracle_cases as OC  # noqa: E402

what = set(sys.argv[1:]) or {"scripted", "catalog"}
if "scripted" in what:
    for W, hp, n, mode, kw in OC.scripted_cases():
        t0 = time.time()
        OC.scripted_ref(W, hp, n, mode, **kw)
        print(f"scripted whisper_full: {n} samples, mode {mode}, {kw}: {time.time() - t0:.1f} s", flush=True)
if "catalog" in what:
    for name in ("small", "medium", "large_v3"):
        t0 = time.time()
        OC.catalog_ref(name)
        print(f"catalog {name}: {time.time() - t0:.1f} s", flush=True)
