"""Register / scratch budget of the RNNoise kernels (cross-compiled here, no GPU needed).

The frame kernel is sized for 4 waves per SIMD (<= 128 VGPRs) and 16 workgroups per CU (<= 10 KB LDS).  Scratch
inside the frame loop is a *correctness* hazard with this compiler (spill stores of a join block are emitted
before its exec restore, see rn_kernels.hip: dotn_h), so the scratch size is pinned: the 12 bytes allowed are
loop-invariant LDS addresses spilled in the prologue under a full exec mask."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_frame_kernel_resource_budget(tmp_path):
    src = os.path.join(ROOT, "crispy_amd", "csrc", "rn_kernels.hip")
    out = subprocess.run(
        [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "rn.o")],
        capture_output=True, text=True, timeout=600, cwd=os.path.dirname(src))
    assert out.returncode == 0, out.stderr[-2000:]
    res = {}
    cur = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    frame = {k: v for k, v in res.items() if "rn_frame_kernel" in k}
    assert len(frame) == 3, list(res)
    for name, r in frame.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 128, (name, r)
        assert r["LDS Size"] <= 10240, (name, r)
        limit = 12 if "ILi0E" in name else 0
        assert r["ScratchSize"] <= limit, (name, r)
