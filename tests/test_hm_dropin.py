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
CFG = os.path.join(ROOT, "tests", "hm", "lowdelay_P_small.cfg")


def _encode(tmp_path, opencl, frames=3, w=192, h=128, extra=()):
    from hmme import synth, yuv
    src = str(tmp_path / "in.yuv")
    pics = []
    for t in range(frames):
        cur, _, _ = synth.make_pair(w, h, seed=5, max_mv=0, noise_sigma=1.0, shift=(2 * t, t), margin=0)
        pics.append(cur.astype(np.uint8))
    yuv.write_luma_420(src, pics)
    env = dict(os.environ, HMME_TRACE="1")
    r = subprocess.run([EXE, "-c", CFG, "-i", src, "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(frames),
                        "-b", str(tmp_path / f"s{opencl}.bin"), f"--OpenCL={opencl}", "--KernelOpenCL=embedded", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pocs = re.findall(r"POC\s+(\d+).*?(\d+) bits \[Y ([0-9.]+) dB", r.stdout)
    return r, [(int(p), int(b), float(y)) for p, b, y in pocs]


def _build():
    if os.path.isdir("/root/reference/source"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "hm-opencl_amd", "csrc")], check=True)
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "ref", "dropin"], check=True)
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/TAppEncoder_hmme not built (needs /root/reference)")


def test_reference_encoder_builds_and_degrades_cleanly_without_gpu(tmp_path):
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r0, p0 = _encode(tmp_path, 0)
    r1, p1 = _encode(tmp_path, 1)
    assert "Create Buffers error" in r1.stdout and "OpenCL Motion Estimation Disabled" in r1.stdout
    assert p0 == p1 and len(p0) == 3          # feature disabled -> identical CPU encode


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
