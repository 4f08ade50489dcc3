"""Register / LDS / scratch budget of the RNNoise kernels (cross-compiled here, no GPU needed).

The frame kernel is sized for 4 waves per SIMD (<= 120 VGPRs, leaving room for a 32-VGPR high-pass wave beside
them) and 16 workgroups per CU (<= 10 KB LDS).
Spills inside the frame loop are a *correctness* hazard with this compiler: VGPR spill stores of a join block
are emitted before the block's exec restore, so a value spilled right after a divergent region is saved for
the active lanes only (see rn_kernels.hip: dotn_h).  Loop-invariant values spilled once in the prologue, under
a full exec mask, are fine -- so the invariant checked on the ISA is "no scratch store after the first loop
header" (reloads inside the loop are harmless)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _makefile_flags():
    """The code-generation flags of the shipped build (crispy_amd/csrc/Makefile: CXXFLAGS), so that what is checked here
    is the ISA that ships: -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize ... (-fPIC / -W* dropped)."""
    text = open(os.path.join(ROOT, "crispy_amd", "csrc", "Makefile")).read()
    m = re.search(r"^CXXFLAGS \?= (.*)$", text, re.M)
    assert m, "CXXFLAGS line not found in the Makefile"
    flags = m.group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f for f in flags if f != "-fPIC" and not f.startswith("-W")]
    assert "-O3" in flags and "--offload-arch=gfx950" in flags and "-fno-slp-vectorize" in flags, flags
    return flags + ["-Wno-unused-function"]


CGFLAGS = _makefile_flags()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("extra", [[], ["-DRN_PROFILE"]], ids=["product", "diagnostic"])
def test_frame_kernel_resource_budget(tmp_path, extra):
    src = os.path.join(ROOT, "crispy_amd", "csrc", "rn_kernels.hip")
    asm = tmp_path / "rn.s"
    out = subprocess.run(
        [HIPCC, *CGFLAGS, "--cuda-device-only", "-S",
         "-Rpass-analysis=kernel-resource-usage", *extra, src, "-o", str(asm)],
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
    assert len(frame) == 2, list(res)      # the fused frame kernel, with and without the diagnostic captures
    text = asm.read_text()
    for name, r in frame.items():
        # 120, not 128: the 32-VGPR high-pass waves must fit as a fifth wave beside four frame waves of a SIMD
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 120, (name, r)
        assert r["LDS Size"] <= 10240, (name, r)
        assert r["ScratchSize"] <= 64, (name, r)
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        loop = body.find("=>This Loop Header: Depth=1")     # the frame loop (the prologue's copy loops are "Inner")
        if loop < 0:                                        # a frame loop without inner loops is labelled "Inner Loop Header"
            loop = body.find("Loop Header: Depth=1")
        assert loop > 0, name
        assert "scratch_store" not in body[loop:], f"{name}: VGPR spill store inside the frame loop"
    # the three-wave stage pipeline (<= 1280 streams): five workgroups per CU need <= 128 registers and <= 32 KB of LDS each
    frame3 = {k: v for k, v in res.items() if "rn_frame3_kernel" in k}
    assert len(frame3) == 2, list(res)
    for name, r in frame3.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 128, (name, r)
        assert r["LDS Size"] <= 32 * 1024, (name, r)
        assert r["ScratchSize"] == 0, (name, r)
    # the high-pass kernels live in their own file (same flags): the 32-register form that fits beside four frame waves
    src_hp = os.path.join(ROOT, "crispy_amd", "csrc", "rn_highpass.hip")
    asm_hp = tmp_path / "rn_hp.s"
    out = subprocess.run(
        [HIPCC, *CGFLAGS, "--cuda-device-only", "-S",
         "-Rpass-analysis=kernel-resource-usage", *extra, src_hp, "-o", str(asm_hp)],
        capture_output=True, text=True, timeout=600, cwd=os.path.dirname(src_hp))
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    text = asm_hp.read_text()
    hp = {k: v for k, v in res.items() if "rn_highpass_kernel" in k}
    assert len(hp) == 2, list(res)        # f32 samples, and the int16 transport of crispy_rn_process_s16*
    deep = {k: v for k, v in res.items() if "rn_highpass_deep_kernel" in k}
    assert len(deep) == 2 and all(r["ScratchSize"] == 0 for r in deep.values()), deep
    for name, r in hp.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 32, (name, r)
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        loop = body.find("Loop Header")
        assert loop > 0, name
        assert "scratch_store" not in body[loop:], f"{name}: VGPR spill store inside the sample loop"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_f16_encoder_kernels_run_at_four_waves_per_simd_without_scratch(tmp_path):
    """whisper_enc_f16.hip: what hides the L2 / HBM round trip of the f16 GEMM is the number of resident waves (a
    second register set of operands in flight measured 18 % slower at three waves), and a spill in an epilogue full
    of predicated stores is the exec-mask hazard described above -- so: <= 128 registers, no scratch, and the
    global -> register requests of a k-block are issued BEFORE its MFMAs (they were once kept in scratch, and once
    sunk below the MFMAs: both cost 2.4x)."""
    src = os.path.join(ROOT, "crispy_amd", "csrc", "whisper_enc_f16.hip")
    asm = tmp_path / "enc.s"
    out = subprocess.run([HIPCC, *CGFLAGS, "--cuda-device-only", "-S",
                          "-Rpass-analysis=kernel-resource-usage", src, "-o", str(asm)],
                         capture_output=True, text=True, timeout=600, cwd=os.path.dirname(src))
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    gem = {k: v for k, v in res.items() if "gemm_hh_kernel" in k}
    assert len(gem) == 4, list(res)      # f16 out, f32 + residual, V^T, conv2 (GELU + positional rows)
    text = asm.read_text()
    for name, r in gem.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 128 and r["ScratchSize"] == 0, (name, r)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]          # (an early-return path has its own s_endpgm)
        loop = body[body.index("Loop Header: Depth=1"):]
        loop = loop[:loop.index("s_barrier")]
        first_load, first_mfma = loop.find("global_load_dwordx4"), loop.find("v_mfma")
        assert 0 < first_load < first_mfma, f"{name}: operand requests are not issued ahead of the MFMAs"
        assert loop.count("global_load_dwordx4") == 4 and loop.count("v_mfma_f32_32x32x16_f16") == 8
    att = {k: v for k, v in res.items() if "attn_enc_h_kernel" in k}
    # (two instances: mode 1's un-normalised probabilities, mode 2's statistics pass + normalised ones)
    assert len(att) == 2 and all(r["ScratchSize"] == 0 and r["VGPRs"] + r.get("AGPRs", 0) <= 128 for r in att.values()), att
    # the LDS-direct main loop (default): three stages of 16 KB, no scratch, and between the barrier and the MFMAs of
    # a k-block nothing but the hand-kept counters: four LDS-DMA requests, eight ds_read_b128, no vmcnt(0)
    hd = {k: v for k, v in res.items() if "gemm_hd_kernel" in k}
    assert len(hd) == 6, list(res)      # + the cross K|V projection (f16 head-major) + the resampler's plain f32 rows
    for name, r in hd.items():
        assert r["ScratchSize"] == 0 and r["LDS Size"] == 3 * 16384 and r["VGPRs"] + r.get("AGPRs", 0) <= 128, (name, r)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]          # (an early-return path has its own s_endpgm)
        loop = body[body.index("Loop Header: Depth=1"):]
        loop = loop[loop.index("s_barrier"):]
        pos, at = [], 0
        for _ in range(8):                                   # up to the eighth MFMA behind the trip's barrier
            at = loop.index("v_mfma_f32_32x32x16_f16", at) + 1
            pos.append(at)
        seg = loop[:pos[-1]]
        assert seg.count("ds_read_b128") == 8 and seg.count("global_load_lds_dwordx4") == 4, name
        assert "vmcnt(0)" not in seg, f"{name}: a compiler-inserted vmcnt(0) serialises the stages"
    # the 256 x 128 and 192 x 128 tile forms (M >= 1024 rows; template argument = epilogue + 8 x row blocks per wave):
    # 72 KB of LDS (the ring or the epilogue images, whichever is larger), two workgroups per CU, MI + 2 LDS-DMA
    # requests, 2 MI + 4 ds_read_b128 and 4 MI MFMAs per trip, again without a compiler-inserted vmcnt(0)
    hd2 = {k: v for k, v in res.items() if "gemm_hd2_kernel" in k}
    assert len(hd2) == 12, list(res)
    for name, r in hd2.items():
        mi = int(re.search(r"gemm_hd2_kernelILi(\d+)E", name).group(1)) >> 3
        assert mi in (3, 4), name
        assert r["ScratchSize"] == 0 and r["LDS Size"] == 3 * 24576 and r["VGPRs"] + r.get("AGPRs", 0) <= 256, (name, r)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        loop = body[body.index("Loop Header: Depth=1"):]
        loop = loop[loop.index("s_barrier"):]
        at = 0
        for _ in range(4 * mi):
            at = loop.index("v_mfma_f32_32x32x16_f16", at) + 1
        seg = loop[:at]
        assert seg.count("ds_read_b128") == 2 * mi + 4 and seg.count("global_load_lds_dwordx4") == mi + 2, name
        assert "vmcnt(0)" not in seg, f"{name}: a compiler-inserted vmcnt(0) serialises the stages"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src_name", ["whisper_kernels.hip", "whisper_dec_f16.hip", "whisper_dec_fused.hip", "mel_kernels.hip",
                                      "resample_kernels.hip"])
def test_asr_kernels_have_no_scratch_at_all(tmp_path, src_name):
    """with this compiler a VGPR spill next to a divergent region is a correctness hazard (see the
    module docstring), and the decode-step kernels -- gemm_skinny_f32_kernel (every LayerNorm / GELU / residual / K-split /
    f16-weight form), attn_dec_kernel, attn_dec_x16_kernel, vocab_f16_kernel, argmax_kernel,
    ts_pick_kernel -- are full of predicated epilogues.  Six skinny forms used to park their epilogue operands in scratch;
    they request them after the K loop now.  The rule for every ASR kernel file: ScratchSize == 0 and no spilled VGPR."""
    src = os.path.join(ROOT, "crispy_amd", "csrc", src_name)
    asm = tmp_path / "k.s"
    out = subprocess.run([HIPCC, *CGFLAGS, "--cuda-device-only", "-S",
                          "-Rpass-analysis=kernel-resource-usage", src, "-o", str(asm)],
                         capture_output=True, text=True, timeout=900, cwd=os.path.dirname(src))
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    assert res, out.stderr[-500:]
    if src_name == "whisper_kernels.hip":
        for must in ("gemm_skinny_f32_kernel", "ts_pick_kernel", "argmax_kernel", "attn_dec_kernel"):
            assert any(must in k for k in res), (must, list(res))
        assert sum("gemm_skinny_f32_kernel" in k for k in res) == 44      # 8 epilogue forms x 4 K splits + 3 x 4 f16-weight forms (residual; bias; bias + GELU)
        assert sum("attn_dec_x16_kernel" in k for k in res) == 5           # 1 / 2 / 4 / 12 key slots per wave + the non-temporal 12 (K|V streams of many clips)
    if src_name == "whisper_dec_fused.hip":       # the fused decode step: every (width, key slots, rows per workgroup) form at <= 128 registers
        for must, n in (("fused_self_kernel", 20), ("fused_cross_kernel", 4), ("fused_mlp_kernel", 8), ("fused_finish_kernel", 2)):
            assert sum(must in k for k in res) == n, (must, [k for k in res if must in k])
        assert all(r["VGPRs"] + r.get("AGPRs", 0) <= 128 for k, r in res.items() if "fused_" in k and "pack" not in k)
    if src_name == "whisper_dec_f16.hip":
        assert sum("vocab_f16_kernel" in k for k in res) == 7              # tiny ... large widths + the two forms that take the final LayerNorm in
    bad = {k: r for k, r in res.items() if r.get("ScratchSize", 0) != 0 or r.get("VGPRs Spill", 0) != 0}
    assert not bad, bad
    assert "scratch_store" not in asm.read_text() and "scratch_load" not in asm.read_text()
