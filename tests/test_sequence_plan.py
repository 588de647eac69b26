"""CPU tests of the sequence driver's host logic (hmme/sequence.py, hmme/synth.py Sequence, hmme/yuv.py LumaFile): launch
batching, the plane-slot replacement plan of the streaming pipeline, the synthetic sequence and the 8 / 16-bit YUV reader."""
import os

import numpy as np
import pytest

from hmme import sequence, shard, synth, yuv


@pytest.mark.parametrize("frames,gop,world,k,slots", [(64, "randomaccess", 1, 1, 8), (64, "randomaccess", 8, 1, 8), (64, "randomaccess", 1, 4, 10),
                                                      (16, "lowdelay_P", 2, 3, 8), (6, "randomaccess", 1, 2, 4), (33, "randomaccess", 3, 16, 34)])
def test_plane_slot_plan_keeps_every_launch_fed(frames, gop, world, k, slots):
    pairs = shard.gop_pairs(frames, gop)
    for rank in range(world):
        mine = [pairs[p] for p in shard.pairs_for_rank(len(pairs), rank, world)]
        batches = sequence.plan_batches(mine, k)
        assert [i for b in batches for i in b] == list(range(len(mine))) and all(1 <= len(b) <= k for b in batches)
        loads, where = sequence.plan_plane_loads(mine, batches, slots)
        assert len(loads) == len(where) == len(batches)
        held = {}                                     # slot -> picture, replayed in issue order
        evict_prev = n_loads = 0
        for b, idx in enumerate(batches):
            need = {p for i in idx for p in mine[i]}
            prev_slots = set(where[b - 1].values()) if b else set()
            for poc, slot in loads[b]:
                assert 0 <= slot < slots
                assert held.get(slot) not in need, "a launch's own picture was evicted for it"
                evict_prev += slot in prev_slots
                n_loads += 1
                held[slot] = poc
            assert set(where[b]) == need
            for p in need:
                assert held[where[b][p]] == p, (b, p)
        distinct = len({p for pr in mine for p in pr})
        assert n_loads >= distinct
        if world == 1 and gop == "randomaccess" and slots >= 8:
            assert n_loads <= distinct + 2, "the random-access working set fits 8 slots: every picture is uploaded once"
            assert evict_prev <= n_loads // 4, "refills should overlap the previous launch, not wait for it"


def test_plane_slot_plan_refuses_too_few_slots():
    pairs = shard.gop_pairs(16, "randomaccess")
    with pytest.raises(ValueError):
        sequence.plan_plane_loads(pairs, sequence.plan_batches(pairs, 4), 3)


@pytest.mark.parametrize("bd", [8, 10])
def test_synthetic_sequence_equals_the_shifted_pair_generator(bd):
    w, h, n = 96, 72, 5
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=bd)
    m = synth.MARGIN
    for t in (0, 2, 4):
        cur, _, _ = synth.make_pair(w, h, seed=777, bit_depth=bd, max_mv=0, noise_sigma=0.0, shift=(3 * t, 2 * t), pad=3 * n + 4)
        assert np.array_equal(seq.luma(t).astype(np.int16), cur[m:m + h, m:m + w])
        assert np.array_equal(seq.padded(t), cur)
        out = np.empty((h, w), np.uint8 if bd == 8 else np.uint16)
        seq.read_into(t, out)
        assert np.array_equal(out, seq.luma(t))
    # picture t+d is picture t displaced by d * (3, 2)
    assert np.array_equal(seq.luma(1)[10:40, 10:40], seq.luma(3)[10 - 4:40 - 4, 10 - 6:40 - 6])


@pytest.mark.parametrize("bd,chroma", [(8, "420"), (10, "420"), (12, "400")])
def test_yuv_file_reader_8_and_16_bit(tmp_path, bd, chroma):
    """TVideoIOYuv::readPlane (TVideoIOYuv.cpp:247): one byte per sample in 8-bit files, two bytes little-endian above"""
    w, h, n = 80, 48, 4
    seq = synth.Sequence(w, h, n, seed=5, bit_depth=bd)
    path = os.path.join(tmp_path, "s.yuv")
    seq.write_yuv(path, chroma)
    fb = 8 if bd == 8 else 16
    assert os.path.getsize(path) == n * yuv.frame_bytes(w, h, fb, chroma)
    f = yuv.LumaFile(path, w, h, fb, chroma)
    assert f.n_frames == n
    buf = np.empty((h, w), np.uint8 if bd == 8 else np.uint16)
    for t in (3, 0, 2):
        f.read_into(t, buf)
        assert np.array_equal(buf, seq.luma(t))
        assert np.array_equal(yuv.read_luma(path, w, h, t, fb, chroma), seq.luma(t))
    with pytest.raises(ValueError):
        f.read_into(n, buf)
    f.close()
    if bd > 8:   # the words are little-endian on disk
        raw = np.fromfile(path, np.uint8, 2)
        assert int(raw[0]) | int(raw[1]) << 8 == int(seq.luma(0)[0, 0])


def test_cpp_sequence_drivers_build_and_keep_the_plan_invariants(tmp_path):
    """hm-opencl_amd/host/SequenceME.{h,cpp} (the sequence driver over the C ABI and the HIP runtime; its planner is the one
    hmme/sequence.py binds) and MultiDeviceME.{h,cpp} (N devices from one process, links RCCL): build, link, and the launch / plane-slot
    plan keeps its invariants (tests/cpp/test_sequence_plan.cpp).  No GPU call is made."""
    import shutil
    import subprocess
    from conftest import ROOT
    if not shutil.which("hipcc"):
        pytest.skip("ROCm headers / libamdhip64 not available")
    host, csrc = os.path.join(ROOT, "hm-opencl_amd", "host"), os.path.join(ROOT, "hm-opencl_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc], check=True)
    subprocess.run(["make", "-s", "-C", host], check=True)
    exe = str(tmp_path / "test_sequence_plan")
    subprocess.run(["g++", "-O1", "-std=c++11", "-pthread", "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_sequence_plan.cpp"),
                    "-L" + host, "-lhmme_multi", "-lhmme_host", "-L" + csrc, "-lhmme", "-Wl,-rpath," + host, "-Wl,-rpath," + csrc,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert os.path.exists(os.path.join(host, "me_stream"))
    # without a GPU the tool fails loudly in hmme_create -- no CPU path behind it
    import torch
    if not torch.cuda.is_available():
        f = tmp_path / "x.yuv"
        f.write_bytes(bytes(64 * 64 * 3 // 2 * 2))
        r = subprocess.run([os.path.join(host, "me_stream"), "--yuv", str(f), "--size", "64x64", "--frames", "2"], capture_output=True, text=True)
        assert r.returncode == 1 and "hmme_create" in r.stderr
        r = subprocess.run([os.path.join(host, "me_stream"), "--yuv", str(f), "--size", "64x64", "--frames", "2", "--gpus", "2"], capture_output=True, text=True)
        assert r.returncode == 1 and "hmme_create on device 0" in r.stderr
