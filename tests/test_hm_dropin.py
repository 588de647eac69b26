"""In-tree drop-in proof.  oracle/Makefile target `dropin` compiles the reference's OWN
encoder sources where they lie, with hm-opencl_amd/host/TEncOpenCL.{h,cpp} + libhmme.so standing in for the
reference's TEncOpenCL.{h,cpp} + libOpenCL (no reference file copied or edited).  The resulting
oracle/_ref/TAppEncoder_hmme is prebuilt here and travels to the GPU box.

CPU (here): the binary builds, and without a GPU `--OpenCL=1` degrades exactly like the reference intends
(createBuffers -> false -> feature disabled -> CPU search; the reference itself segfaults in findDevice).
GPU: `--OpenCL=1` routes every 64x64 2Nx2N integer search through the HIP engine."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_hmme")
EXE_HM = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_hmme_hm")   # + tools/hm_patch (CPU-exact mode, bi-pred tables, edge CTUs)
CFG = os.path.join(ROOT, "tests", "hm", "lowdelay_P_small.cfg")
CFG_B = os.path.join(ROOT, "tests", "hm", "lowdelay_B_small.cfg")
CFG_P4 = os.path.join(ROOT, "tests", "hm", "lowdelay_P4_small.cfg")      # four active references (refIdx 0..3)
CFG_RA = os.path.join(ROOT, "tests", "hm", "randomaccess_small.cfg")     # hierarchical B, GOP 8, references on both sides


def _encode(tmp_path, opencl, frames=3, w=192, h=128, extra=(), exe=None, cfg=None, env_extra=None, fade=0.0):
    from hmme import synth, yuv
    src = str(tmp_path / "in.yuv")
    pics = []
    for t in range(frames):
        cur, _, _ = synth.make_pair(w, h, seed=5, max_mv=0, noise_sigma=1.0, shift=(2 * t, t), margin=0)
        if fade:   # a fade to black: what explicit weighted prediction is for
            cur = np.clip(np.rint(cur * (1.0 - fade * t)), 0, 255)
        pics.append(cur.astype(np.uint8))
    yuv.write_luma_420(src, pics)
    env = dict(os.environ, HMME_TRACE="1", **(env_extra or {}))
    r = subprocess.run([exe or EXE, "-c", cfg or CFG, "-i", src, "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(frames),
                        "-b", str(tmp_path / f"s{opencl}.bin"), f"--OpenCL={opencl}", "--KernelOpenCL=embedded", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pocs = re.findall(r"POC\s+(\d+).*?(\d+) bits \[Y ([0-9.]+) dB", r.stdout)
    return r, [(int(p), int(b), float(y)) for p, b, y in pocs]


def _build():
    if os.path.isdir("/root/reference/source"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "hm-opencl_amd", "csrc")], check=True)
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "ref", "dropin"], check=True)
    if not (os.path.exists(EXE) and os.path.exists(EXE_HM)):
        pytest.skip("oracle/_ref/TAppEncoder_hmme{,_hm} not built (needs /root/reference)")


def test_reference_encoder_builds_and_degrades_cleanly_without_gpu(tmp_path):
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r0, p0 = _encode(tmp_path, 0)
    r1, p1 = _encode(tmp_path, 1)
    assert "Create Buffers error" in r1.stdout and "OpenCL Motion Estimation Disabled" in r1.stdout
    assert p0 == p1 and len(p0) == 3          # feature disabled -> identical CPU encode
    r2, p2 = _encode(tmp_path, 1, exe=EXE_HM)  # the patched encoder degrades the same way: the inserted block is never entered
    assert "OpenCL Motion Estimation Disabled" in r2.stdout and p2 == p0


@pytest.mark.gpu
def test_reference_encoder_runs_on_the_hip_engine(tmp_path):
    _build()
    r0, p0 = _encode(tmp_path, 0)
    r1, p1 = _encode(tmp_path, 1)
    assert "Buffers created" in r1.stdout
    m = re.search(r"TEncOpenCL\(hmme\): (\d+) calcMotionVectors calls, (\d+) failed", r1.stderr)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) == 0, r1.stderr[-1000:]
    assert len(p1) == 3 and p1[0] == p0[0]                      # intra picture unaffected
    for (_, b0, y0), (_, b1, y1) in zip(p0[1:], p1[1:]):         # exhaustive GPU search vs TZ: same ballpark
        assert abs(y1 - y0) < 1.5 and b1 < 2 * b0 + 2000
    print("OpenCL=0:", p0, "\\nOpenCL=1 (hmme):", p1, "\\n", m.group(0))


_TRACE = re.compile(r"TEncOpenCL\(hmme\): (\d+) calcMotionVectors calls, (\d+) failed, (\d+) edge-CTU, (\d+) bi-pred, (\d+) results verified "
                    r"against xPatternSearch, (\d+) differ")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,bipred,frames", [(CFG, False, 4), (CFG_B, True, 4), (CFG_P4, False, 6), (CFG_RA, True, 9)])
def test_patched_encoder_hm_mode_equals_hm_cpu_search(tmp_path, cfg, bipred, frames):
    """tools/hm_patch applied (oracle/_ref/TAppEncoder_hmme_hm), 208x120 = 4 x 2 CTUs of which only 3 are whole.  HMME_VERIFY=1 makes
    the encoder run HM's OWN xPatternSearch beside every engine call: the 64x64 2Nx2N PU and three more slots per call must
    come out identical (MV and ruiCost) -- i.e. what --FastSearch=0 computes for those PUs on the same inputs -- for uni- and
    bi-prediction calls.  Edge CTUs are searched (not served from the previous CTU's tables), bi-prediction uses its own tables."""
    _build()
    r, p = _encode(tmp_path, 1, frames=frames, w=208, h=120, exe=EXE_HM, cfg=cfg, env_extra={"HMME_VERIFY": "1"}, extra=("--SearchRange=24",))
    m = _TRACE.search(r.stderr)
    assert m, r.stderr[-1500:]
    calls, failed, edge, bi, verified, differ = (int(v) for v in m.groups())
    assert failed == 0 and differ == 0, m.group(0)
    assert calls > 0 and edge > 0 and verified >= 4 * (calls - edge - bi)
    assert (bi > 0) == bipred, m.group(0)
    assert len(p) == frames
    # the same binary with HMME_HM_MODE=0 is the unmodified reference call sequence
    r0, p0 = _encode(tmp_path, 1, frames=frames, w=208, h=120, exe=EXE_HM, cfg=cfg, env_extra={"HMME_HM_MODE": "0"}, extra=("--SearchRange=24",))
    rc, pc = _encode(tmp_path, 1, frames=frames, w=208, h=120, exe=EXE, cfg=cfg, extra=("--SearchRange=24",))
    assert p0 == pc and int(_TRACE.search(r0.stderr).group(3)) == 0
    # against HM's own searches: exhaustive-search quality (bits within a few per cent of --FastSearch=0)
    rf, pf = _encode(tmp_path, 0, frames=frames, w=208, h=120, exe=EXE_HM, cfg=cfg, extra=("--SearchRange=24", "--FastSearch=0"))
    bits, bits_full = sum(b for _, b, _ in p[1:]), sum(b for _, b, _ in pf[1:])
    assert bits < 1.06 * bits_full + 500, (bits, bits_full)
    print("hm mode:", p, "\nFastSearch=0:", pf, "\n", m.group(0))


@pytest.mark.gpu
def test_patched_encoder_gpu_refinement_tables(tmp_path):
    """HMME_GPU_FRAC=1: xPatternSearchFracDIF (TEncSearch.cpp:3798) served from the tables hmme_search_refine_ctu fills in the
    same engine call as the integer search.  HMME_VERIFY=1 additionally runs HM's own refinement for every 64x64 2Nx2N PU: the
    quarter-pel MV and ruiCost from the table must be identical.  Smaller PUs price the MV against their own predictor."""
    _build()
    common = dict(frames=4, w=208, h=120, exe=EXE_HM, cfg=CFG, extra=("--SearchRange=24",))
    r, p = _encode(tmp_path, 1, env_extra={"HMME_VERIFY": "1", "HMME_GPU_FRAC": "1"}, **common)
    m = _TRACE.search(r.stderr)
    assert m, r.stderr[-1500:]
    calls, failed, edge, bi, verified, differ = (int(v) for v in m.groups())
    assert failed == 0 and differ == 0 and calls > 0, m.group(0)
    r1, p1 = _encode(tmp_path, 1, env_extra={"HMME_VERIFY": "1"}, **common)
    assert verified > int(_TRACE.search(r1.stderr).group(5))          # the refinement checks came on top of the integer ones
    bits, bits_cpu_frac = sum(b for _, b, _ in p[1:]), sum(b for _, b, _ in p1[1:])
    assert abs(bits - bits_cpu_frac) < 0.05 * bits_cpu_frac + 300, (bits, bits_cpu_frac)
    assert all(abs(a[2] - b[2]) < 0.5 for a, b in zip(p, p1))
    print("GPU refinement tables:", p, "\nCPU xPatternSearchFracDIF:", p1, "\n", m.group(0))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,frames,extra", [(CFG_B, 4, ()), (CFG_RA, 9, ()), (CFG_B, 3, ("--Profile=main10", "--InternalBitDepth=10"))])
def test_patched_encoder_gpu_refinement_of_the_biprediction_pass(tmp_path, cfg, frames, extra):
    """HMME_GPU_FRAC=1 in B pictures: the bBi pass runs xPatternSearchFracDIF on the origin 2*org - pred_other (TEncSearch.cpp:3702-3712,
    :3798); the engine refines that origin in the same call as its integer search (hmme_search_refine_ctu on samples outside the
    range) and the patch serves the 64x64 PU from those tables.  HMME_VERIFY=1 runs HM's OWN xPatternSearchFracDIF(..., bBi) beside
    every such lookup: quarter-pel MV and ruiCost must be identical, for every bi-prediction call, 8 and 10 bit."""
    _build()
    common = dict(frames=frames, w=208, h=120, exe=EXE_HM, cfg=cfg, extra=("--SearchRange=24",) + extra)
    r, p = _encode(tmp_path, 1, env_extra={"HMME_VERIFY": "1", "HMME_GPU_FRAC": "1"}, **common)
    m = _TRACE.search(r.stderr)
    assert m, r.stderr[-1500:]
    calls, failed, edge, bi, verified, differ = (int(v) for v in m.groups())
    assert failed == 0 and differ == 0 and bi > 0, m.group(0)
    r1, p1 = _encode(tmp_path, 1, env_extra={"HMME_VERIFY": "1"}, **common)
    m1 = _TRACE.search(r1.stderr)
    calls1, _, edge1, bi1, verified1, differ1 = (int(v) for v in m1.groups())
    assert differ1 == 0
    # with the tables on, every whole-CTU call adds a refinement check: the uni-prediction ones AND the bi-prediction ones (the
    # encoders may take different decisions later on, so compare per-call rates, not totals)
    assert verified - 4 * (calls - edge - bi) - bi >= (calls - edge - bi) + bi - 2, (m.group(0), m1.group(0))
    bits, bits_cpu_frac = sum(b for _, b, _ in p[1:]), sum(b for _, b, _ in p1[1:])
    assert abs(bits - bits_cpu_frac) < 0.05 * bits_cpu_frac + 300, (bits, bits_cpu_frac)
    print("bi-prediction refinement from the tables:", p, "\nCPU xPatternSearchFracDIF:", p1, "\n", m.group(0))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,wp_flag,frames", [(CFG, "--WeightedPredP=1", 4), (CFG_B, "--WeightedPredB=1", 4)])
def test_patched_encoder_searches_weighted_prediction_slices_on_the_engine(tmp_path, cfg, wp_flag, frames):
    """--WeightedPredP/B=1 on a fade: HM estimates explicit weights, and its integer search then prices weighted SADs (xGetSADw,
    setWpScalingDistParam at TEncSearch.cpp:3740).  Round 4: the patch hands the slice's luma weights to the engine (setWeight ->
    hmme_search_ctu_w) instead of sending such slices back to the CPU search.  HMME_VERIFY=1 runs HM's OWN xPatternSearch -- with
    bApplyWeight, i.e. xGetSADw -- beside every engine call: zero differences, on at least as many engine calls as the unweighted
    encode of the same clip makes, and exhaustive-search quality against --FastSearch=0."""
    _build()
    common = dict(frames=frames, w=208, h=120, cfg=cfg, extra=("--SearchRange=24", wp_flag), fade=0.12)
    r, p = _encode(tmp_path, 1, exe=EXE_HM, env_extra={"HMME_VERIFY": "1"}, **common)
    m = _TRACE.search(r.stderr)
    assert m, r.stderr[-1500:]
    calls, failed, edge, bi, verified, differ = (int(v) for v in m.groups())
    assert failed == 0 and differ == 0, m.group(0)
    assert calls > 0 and verified >= 4 * (calls - edge - bi), m.group(0)
    assert "weighted prediction" not in r.stderr                       # no call was refused
    # the slices really carried weights: the same encode without the flag produces another bitstream
    rn, pn = _encode(tmp_path, 1, exe=EXE_HM, **dict(common, extra=("--SearchRange=24",)))
    assert pn != p
    mn = _TRACE.search(rn.stderr)
    assert mn and calls >= int(mn.group(1)) // 2, (m.group(0), mn.group(0))   # weighted slices are searched on the engine, not skipped
    rf, pf = _encode(tmp_path, 0, exe=EXE_HM, **dict(common, extra=("--SearchRange=24", wp_flag, "--FastSearch=0")))
    bits, bits_full = sum(b for _, b, _ in p[1:]), sum(b for _, b, _ in pf[1:])
    assert bits < 1.06 * bits_full + 500, (bits, bits_full)
    # ... and with the refinement tables: the weighted xPatternSearchFracDIF of the 64x64 PU comes from the engine (xGetHADsw pricing),
    # HM's own weighted refinement runs beside every lookup: more checks than before, still none differs
    rg, pg = _encode(tmp_path, 1, exe=EXE_HM, env_extra={"HMME_VERIFY": "1", "HMME_GPU_FRAC": "1"}, **common)
    mg = _TRACE.search(rg.stderr)
    assert mg, rg.stderr[-1500:]
    calls_g, failed_g, _, _, verified_g, differ_g = (int(v) for v in mg.groups())
    assert failed_g == 0 and differ_g == 0 and verified_g > verified, (mg.group(0), m.group(0))
    print("WP fade on the engine:", p, "\nFastSearch=0:", pf, "\n", m.group(0), "\nwith refinement tables:", mg.group(0))


@pytest.mark.gpu
def test_encoders_at_10_bit_internal_depth(tmp_path):
    """InternalBitDepth 10 (8-bit input, 10-bit coding): the unmodified call sites cannot tell the class the bit depth -- it takes the
    sample width from the reference window and, like cl/sad.cl, leaves the sums unshifted; no call may fail or serve stale tables.
    The patched encoder passes the SPS bit depth (16-bit kernels, HM's >> 2): HM's own xPatternSearch must agree on every check."""
    _build()
    common = dict(frames=3, w=208, h=120, cfg=CFG, extra=("--SearchRange=16", "--Profile=main10", "--InternalBitDepth=10"))
    r, p = _encode(tmp_path, 1, exe=EXE, **common)
    m = _TRACE.search(r.stderr)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) == 0, r.stderr[-1500:]
    r2, p2 = _encode(tmp_path, 1, exe=EXE_HM, env_extra={"HMME_VERIFY": "1", "HMME_GPU_FRAC": "1"}, **common)
    m2 = _TRACE.search(r2.stderr)
    calls, failed, edge, bi, verified, differ = (int(v) for v in m2.groups())
    assert failed == 0 and differ == 0 and verified > 4 * (calls - edge), m2.group(0)
    r0, p0 = _encode(tmp_path, 0, exe=EXE_HM, **common)
    assert len(p) == len(p2) == len(p0) == 3
    for (_, b0, y0), (_, b1, y1), (_, b2, y2) in zip(p0[1:], p[1:], p2[1:]):
        assert abs(y1 - y0) < 1.5 and abs(y2 - y0) < 1.0 and b2 < 1.3 * b0 + 1000
    print("10-bit: TZ", p0, "\ncompat", p, "\nhm mode", p2, m2.group(0))


def test_hm_patch_is_a_pure_insertion_and_applies(tmp_path):
    """tools/hm_patch/TEncSearch_hm_mode.patch: no reference line removed (nothing of the reference is restated beyond one line of
    context per hunk edge), and it applies cleanly to the reference file where that is present"""
    import subprocess
    patch = os.path.join(ROOT, "tools", "hm_patch", "TEncSearch_hm_mode.patch")
    lines = open(patch).read().splitlines()
    body = [ln for ln in lines if not ln.startswith(("---", "+++", "@@"))]
    assert not [ln for ln in body if ln.startswith("-")], "the patch removes reference lines"
    assert sum(1 for ln in body if ln.startswith("+")) > 100
    assert sum(1 for ln in body if ln.startswith(" ")) <= 2 * sum(1 for ln in lines if ln.startswith("@@"))
    ref = "/root/reference/source/Lib/TLibEncoder/TEncSearch.cpp"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    out = tmp_path / "TEncSearch.cpp"
    r = subprocess.run(["patch", "-s", "-o", str(out), ref, patch], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    n_ref, n_new = len(open(ref).read().splitlines()), len(open(out).read().splitlines())
    assert n_new - n_ref == sum(1 for ln in body if ln.startswith("+"))
