"""GPU tests of the sequence path (BASELINE.json config 4): picture pairs of a GOP searched pair-sharded, several pairs per
launch, streamed through a ring of planes, gathered with RCCL -- every table that comes out compared with the CPU oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TOOL = os.path.join(ROOT, "tools", "me_sequence.py")


@pytest.fixture(scope="module")
def engine():
    from hmme import api
    e = api.Engine(0, 128)
    yield e
    e.close()


def _run_tool(args, torchrun=False, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    if torchrun:
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), TOOL] + args
    else:
        cmd = [sys.executable, TOOL] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def _check_dump_against_oracle(oracle_lib, dump, seq, w, h, sr, bd, lq, n_threads=16):
    """every dumped (pair, CTU, slot) against the oracle's exhaustive search of that picture pair"""
    from hmme import synth
    d = np.load(dump)
    m = synth.MARGIN
    c0, cn = int(d["ctu_first"]), d["mv"].shape[1]
    padded = {}
    for i, (cur, ref) in enumerate(d["pairs"].tolist()):
        for t in (cur, ref):
            if t not in padded:
                padded[t] = seq.padded(t)
        ox, oy, osad = oracle_lib.search_frame(padded[cur], padded[ref], (m, m), w, h, sr, None, lq, 1, bd, c0, cn, n_threads)
        mv, sad = d["mv"][i], d["sad"][i]
        bad = np.flatnonzero((mv[:, :, 0] != ox).any(axis=1) | (mv[:, :, 1] != oy).any(axis=1) | (sad != osad).any(axis=1))
        assert bad.size == 0, f"pair {d['pair_index'][i]} ({cur}, {ref}): CTUs {bad[:10] + c0} differ from the oracle"
    return d


@pytest.mark.parametrize("mode", ["resident", "stream", "torchrun_rccl"])
def test_config4_small_gathered_tables_vs_oracle(tmp_path, oracle_lib, mode):
    """BASELINE config 4 at 640x448 x 6 pictures: the tables that come out of shard.gather_pair_results -- every pair of the
    random-access GOP (cfg/encoder_randomaccess_main.cfg:28-31), every CTU, all 593 slots -- against the oracle.  `stream`:
    pictures through a 4-slot plane ring, three pairs per launch, tables downloaded on a third stream; `torchrun_rccl`: the same
    run under torchrun with one rank, the tables carried by all_gather_into_tensor on RCCL."""
    from hmme import shard, synth
    w, h, n, sr = 640, 448, 6, 16
    dump = os.path.join(tmp_path, "t.npz")
    args = ["--frames", str(n), "--gop", "randomaccess", "--size", f"{w}x{h}", "--search-range", str(sr), "--dump", dump]
    if mode == "stream":
        args += ["--stream", "--slots", "5", "--pairs-per-launch", "3", "--download"]
    d = _run_tool(args, torchrun=(mode == "torchrun_rccl"), env={"HMME_SEQ_FORCE_DIST": "1"} if mode == "torchrun_rccl" else None)
    pairs = shard.gop_pairs(n, "randomaccess")
    assert d["pairs"] == len(pairs) == 9 and d["pair_list"] == [list(p) for p in pairs]
    if mode == "torchrun_rccl":
        assert "nccl" in d["mode"]["collective"]
    if mode == "stream":
        assert d["rank0"]["launches"] == 3 and d["rank0"]["plane_slots"] <= 5 and d["rank0"]["uploads"] >= 6
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=8)
    out = _check_dump_against_oracle(oracle_lib, dump, seq, w, h, sr, 8, lq)
    assert out["mv"].shape == (9, 70, 593, 2)
    if mode == "stream":
        assert bool(out["host_equal"]), "the tables the download stream delivered differ from the device tables"
    # picture t is the texture displaced by (3t, 2t): the 64x64 winner of pair (cur, ref) is (3, 2) * (cur - ref)
    for (cur, ref), med in zip(pairs, d["median_mv_64x64"]):
        assert med == [3 * (cur - ref), 2 * (cur - ref)]


def test_config4_full_size_2160p_64_pictures(tmp_path, oracle_lib):
    """BASELINE config 4 itself on one GPU: 3840x2160, 64 pictures, random-access GOP = 124 picture pairs, SR 64.  Every pair's
    median 64x64 MV is the planted displacement; one CTU row of three pairs (the first, a B picture with a forward reference,
    the last) equals the oracle in all 593 slots."""
    from hmme import shard, synth
    w, h, n, sr = 3840, 2160, 64, 64
    pairs = shard.gop_pairs(n, "randomaccess")
    assert len(pairs) == 124
    fwd = next(i for i, (c, r) in enumerate(pairs) if r - c == 3 and c > 32)      # a B picture POC 4k+1 referencing 4k+4 (+3)
    sel = [0, fwd, len(pairs) - 1]
    dump = os.path.join(tmp_path, "t.npz")
    d = _run_tool(["--frames", str(n), "--gop", "randomaccess", "--size", "2160p", "--search-range", str(sr), "--dump", dump,
                   "--dump-pairs", ",".join(str(i) for i in sel), "--dump-ctus", f"{60 * 17}:60"])
    assert d["pairs"] == 124 and d["gpus"] == 1 and d["rank0"]["launches"] == 124
    for (cur, ref), med in zip(pairs, d["median_mv_64x64"]):
        assert med == [3 * (cur - ref), 2 * (cur - ref)], (cur, ref, med)
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=8)
    out = _check_dump_against_oracle(oracle_lib, dump, seq, w, h, sr, 8, lq)
    assert out["mv"].shape == (3, 60, 593, 2) and pairs[fwd][1] - pairs[fwd][0] == 3
    assert d["pairs_per_s"] > 100      # ~2.6 ms per pair resident; a generous floor that still catches a serialised path


@pytest.mark.parametrize("bd,w,h,n,sr,k,slots", [(8, 832, 480, 10, 32, 2, 4), (10, 640, 360, 6, 24, 1, 2)])
def test_streamed_yuv_file_equals_resident_equals_oracle(tmp_path, oracle_lib, bd, w, h, n, sr, k, slots):
    """the frame feeder end to end: an 8-bit and a 16-bit little-endian YUV file (TVideoIOYuv.cpp:247) read by
    the reader thread into page-locked buffers, uploaded on the copy stream into a plane ring smaller than the sequence (the
    two-slot ring of the 10-bit case re-uploads pictures: every refill there evicts a plane the previous launch read), searched
    k pairs per launch and refined on the compute stream, tables downloaded on a third -- equal to the resident run of the same
    file, and to the oracle for every pair, CTU and slot."""
    from hmme import sequence, shard, synth
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=bd)
    path = os.path.join(tmp_path, "seq.yuv")
    seq.write_yuv(path)
    base = ["--frames", str(n), "--gop", "randomaccess", "--size", f"{w}x{h}", "--search-range", str(sr), "--bit-depth", str(bd), "--yuv", path,
            "--refine"]
    d_res, d_str = os.path.join(tmp_path, "res.npz"), os.path.join(tmp_path, "str.npz")
    r1 = _run_tool(base + ["--dump", d_res])
    r2 = _run_tool(base + ["--dump", d_str, "--stream", "--slots", str(slots), "--pairs-per-launch", str(k), "--download"])
    assert r1["mode"]["stream"] is False and r2["mode"]["stream"] is True and r2["rank0"]["plane_slots"] == slots
    pairs = shard.gop_pairs(n, "randomaccess")
    loads, _ = sequence.plan_plane_loads(pairs, sequence.plan_batches(pairs, k), slots)
    assert r2["rank0"]["uploads"] == sum(len(l) for l in loads) and r1["rank0"]["uploads"] == n
    if slots == 2:
        assert r2["rank0"]["uploads"] > n, "a two-slot ring cannot hold the GOP's working set: pictures must have been re-uploaded"
    a, b = np.load(d_res), np.load(d_str)
    for k in ("mv", "sad", "qmv", "cost"):
        assert np.array_equal(a[k], b[k]), f"streamed {k} tables differ from the resident ones"
    assert bool(b["host_equal"])
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    out = _check_dump_against_oracle(oracle_lib, d_str, seq, w, h, sr, bd, lq)
    # refinement of the streamed run against the oracle's xPatternSearchFracDIF, first and last pair
    m = synth.MARGIN
    for i in (0, out["mv"].shape[0] - 1):
        cur, ref = (int(v) for v in out["pairs"][i])
        oq, oc = oracle_lib.refine_frame(seq.padded(cur), seq.padded(ref), (m, m), w, h, out["mv"][i], None, lq, 1, bd, n_threads=16)
        assert np.array_equal(out["qmv"][i], oq) and np.array_equal(out["cost"][i], oc)


@pytest.mark.parametrize("bd,w,h,n,sr,k,slots,gop", [(8, 640, 448, 6, 16, 1, 0, "randomaccess"), (8, 832, 480, 10, 32, 3, 5, "randomaccess"),
                                                    (10, 640, 360, 6, 24, 2, 4, "randomaccess"), (8, 320, 192, 7, 16, 4, 0, "lowdelay_P")])
def test_cpp_sequence_driver_streams_a_yuv_file_equal_to_the_oracle(tmp_path, oracle_lib, bd, w, h, n, sr, k, slots, gop):
    """tools/me_stream.cpp over hm-opencl_amd/host/SequenceME (C++: reader thread, page-locked buffers, copy / compute / download
    streams, the C ABI and the HIP runtime only -- no Python, no torch in the process): tables of every pair, CTU and slot of a
    streamed 8- / 16-bit YUV file, search and refinement, against the oracle; the pair list equals hmme.shard.gop_pairs."""
    from conftest import ROOT
    from hmme import shard, synth
    host = os.path.join(ROOT, "hm-opencl_amd", "host")
    subprocess.run(["make", "-s", "-C", host], check=True)
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=bd)
    path, out = os.path.join(tmp_path, "seq.yuv"), os.path.join(tmp_path, "tables.bin")
    seq.write_yuv(path)
    r = subprocess.run([os.path.join(host, "me_stream"), "--yuv", path, "--size", f"{w}x{h}", "--frames", str(n), "--gop", gop, "--search-range", str(sr),
                        "--bit-depth", str(bd), "--pairs-per-launch", str(k), "--slots", str(slots), "--refine", "--out", out, "--repeat", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    pairs = shard.gop_pairs(n, gop)
    raw = np.fromfile(out, np.int32)
    n_pairs, n_ctu, refined = (int(v) for v in raw[:3])
    assert n_pairs == len(pairs) == d["pairs"] and n_ctu == ((w + 63) // 64) * ((h + 63) // 64) and refined == 1
    got_pairs = raw[4:4 + 2 * n_pairs].reshape(n_pairs, 2)
    assert got_pairs.tolist() == [list(p) for p in pairs]
    body = raw[4 + 2 * n_pairs:]
    per = n_pairs * n_ctu * 593
    mv = body[:per].view(np.int16).reshape(n_pairs, n_ctu, 593, 2)
    sad = body[per:2 * per].view(np.uint32).reshape(n_pairs, n_ctu, 593)
    qmv = body[2 * per:3 * per].view(np.int16).reshape(n_pairs, n_ctu, 593, 2)
    cost = body[3 * per:4 * per].view(np.uint32).reshape(n_pairs, n_ctu, 593)
    assert d["launches"] == (n_pairs + k - 1) // k and d["uploads"] >= len({p for pr in pairs for p in pr})
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    m = synth.MARGIN
    padded = {t: seq.padded(t) for t in {p for pr in pairs for p in pr}}
    for i, (cur, ref) in enumerate(pairs):
        ox, oy, osad = oracle_lib.search_frame(padded[cur], padded[ref], (m, m), w, h, sr, None, lq, 1, bd, n_threads=16)
        assert np.array_equal(mv[i, :, :, 0], ox) and np.array_equal(mv[i, :, :, 1], oy) and np.array_equal(sad[i], osad), (i, cur, ref)
        if i in (0, n_pairs - 1):
            oq, oc = oracle_lib.refine_frame(padded[cur], padded[ref], (m, m), w, h, mv[i], None, lq, 1, bd, n_threads=16)
            assert np.array_equal(qmv[i], oq) and np.array_equal(cost[i], oc), (i, cur, ref)


def _read_tables(path):
    raw = np.fromfile(path, np.int32)
    n_pairs, n_ctu, refined = (int(v) for v in raw[:3])
    body = raw[4 + 2 * n_pairs:]
    per = n_pairs * n_ctu * 593
    t = {"pairs": raw[4:4 + 2 * n_pairs].reshape(n_pairs, 2).tolist(), "mv": body[:per].view(np.int16).reshape(n_pairs, n_ctu, 593, 2),
         "sad": body[per:2 * per].view(np.uint32).reshape(n_pairs, n_ctu, 593)}
    if refined:
        t["qmv"] = body[2 * per:3 * per].view(np.int16).reshape(n_pairs, n_ctu, 593, 2)
        t["cost"] = body[3 * per:4 * per].view(np.uint32).reshape(n_pairs, n_ctu, 593)
    return t


@pytest.mark.parametrize("devices,gather", [("0,0", "peer"), ("0,0,0", "host"), ("0", "rccl"), ("0,0", "rccl")])
def test_cpp_multi_device_driver_on_contexts_sharing_the_one_gpu(tmp_path, oracle_lib, devices, gather):
    """hm-opencl_amd/host/MultiDeviceME (C++, one process, one host thread + hmme context + sequence driver per rank, pair p -> rank
    p mod N, tables gathered into rank 0's page-locked memory in pair order) rehearsed on this one-GPU box with N contexts on device
    0: the gathered tables equal the single-context driver's, and the oracle's.  `peer` = hipMemcpyPeerAsync into device 0's gather
    buffer, `host` = every rank downloads into its places, `rccl` = ncclCommInitAll + grouped ncclSend / ncclRecv -- which exists for
    N DISTINCT devices (unmeasured here: one GPU), is exercised with world 1, and must REFUSE one GPU twice."""
    from conftest import ROOT
    from hmme import shard, synth
    host = os.path.join(ROOT, "hm-opencl_amd", "host")
    subprocess.run(["make", "-s", "-C", host], check=True)
    w, h, n, sr = 448, 256, 9, 24
    seq = synth.Sequence(w, h, n, seed=777, bit_depth=8)
    path = os.path.join(tmp_path, "seq.yuv")
    seq.write_yuv(path)
    common = ["--yuv", path, "--size", f"{w}x{h}", "--frames", str(n), "--gop", "randomaccess", "--search-range", str(sr), "--pairs-per-launch", "2", "--refine"]
    out_m, out_1 = os.path.join(tmp_path, "multi.bin"), os.path.join(tmp_path, "single.bin")
    r = subprocess.run([os.path.join(host, "me_stream")] + common + ["--devices", devices, "--gather", gather, "--out", out_m, "--repeat", "2"],
                       capture_output=True, text=True, timeout=600)
    world = len(devices.split(","))
    if gather == "rccl" and world > 1:
        assert r.returncode != 0 and "distinct devices" in r.stderr
        return
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    pairs = shard.gop_pairs(n, "randomaccess")
    n_ctu = ((w + 63) // 64) * ((h + 63) // 64)
    assert d["gpus"] == world and d["pairs"] == len(pairs) and d["pairs_per_device"] == shard.pair_counts(len(pairs), world)
    assert d["gather_bytes"] == sum(shard.pair_counts(len(pairs), world)[1:]) * n_ctu * 593 * 4 * 4   # four tables of the OTHER ranks' pairs
    r1 = subprocess.run([os.path.join(host, "me_stream")] + common + ["--out", out_1], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    tm, t1 = _read_tables(out_m), _read_tables(out_1)
    assert tm["pairs"] == t1["pairs"] == [list(p) for p in pairs]
    for k in ("mv", "sad", "qmv", "cost"):
        assert np.array_equal(tm[k], t1[k]), k
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    m = synth.MARGIN
    for i in (0, 1, len(pairs) - 2, len(pairs) - 1):      # pairs of both / all ranks
        cur, ref = pairs[i]
        ox, oy, osad = oracle_lib.search_frame(seq.padded(cur), seq.padded(ref), (m, m), w, h, sr, None, lq, 1, 8, n_threads=16)
        assert np.array_equal(tm["mv"][i, :, :, 0], ox) and np.array_equal(tm["mv"][i, :, :, 1], oy) and np.array_equal(tm["sad"][i], osad), (i, cur, ref)


@pytest.mark.parametrize("w,h,n,sr,bd", [(480, 320, 12, 16, 8), (1920, 1080, 8, 64, 8), (384, 256, 9, 24, 10)])
def test_fuzz_streaming_configurations_equal_the_resident_run(engine, w, h, n, sr, bd):
    """the streaming pipeline under many shapes -- plane rings from the bare minimum (every refill evicts a plane that was just read) to
    roomy, 1..4 pairs per launch, 1..4 host buffers, refinement / download on and off -- must give the tables of the resident run, bit
    for bit: a refill that overtook a search, a host buffer reused too early or a launch that read a plane before its fill would show"""
    import torch
    from hmme import sequence, shard, synth
    engine.set_lambda(57.9)
    seq = synth.Sequence(w, h, n, seed=99, bit_depth=bd)
    pairs = shard.gop_pairs(n, "randomaccess")
    dev = torch.device("cuda", 0)
    base = sequence.run_rank(engine, seq, pairs, w, h, bd, sr, refine=True, device=dev)
    want = {k: base[k].cpu() for k in ("mv", "sad", "qmv", "cost")}
    rng = np.random.default_rng(w + n)
    n_cfg = int(os.environ.get("HMME_STREAM_FUZZ_CASES", "10" if w < 1000 else "4"))
    keep = sequence.RankResources()      # every other pass runs on plane slots / host buffers / streams kept from earlier passes (what holds a ring's
    for it in range(n_cfg):              # planes from the pass before is stale and must be overwritten before it is read)
        k = int(rng.integers(1, 5))
        need = max(len({p for i in b for p in pairs[i]}) for b in sequence.plan_batches(pairs, k))
        slots = int(rng.integers(need, need + 4)) if it % 3 else need            # every third run on the smallest possible ring
        refine, download, hb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(1, 5))
        got = sequence.run_rank(engine, seq, pairs, w, h, bd, sr, stream_mode=True, pairs_per_launch=k, refine=refine, download=download,
                                n_slots=slots, device=dev, host_buffers=hb, resources=keep if it % 2 else None)
        tag = dict(k=k, slots=slots, refine=refine, download=download, host_buffers=hb, kept=bool(it % 2))
        for name in ("mv", "sad") + (("qmv", "cost") if refine else ()):
            assert torch.equal(got[name].cpu(), want[name]), (tag, name)
            if download:
                assert torch.equal(got["host_" + name], want[name]), (tag, "host_" + name)
        assert got["plane_slots"] <= slots and got["uploads"] >= len({p for pr in pairs for p in pr})
    # the same geometry twice on kept resources, the second time a share of the pairs only (bench.py config4_sharded: rank 0 alone, then its share)
    for share in (pairs, pairs[1::2]):
        got = sequence.run_rank(engine, seq, share, w, h, bd, sr, stream_mode=True, device=dev, resources=keep)
        idx = [pairs.index(p) for p in share]
        assert torch.equal(got["mv"].cpu(), want["mv"][idx]) and torch.equal(got["sad"].cpu(), want["sad"][idx])
    assert keep.planes and keep.key is not None
    keep.close()
    assert not keep.planes


def _planes(engine, w, h, bd, imgs):
    pls = []
    for img in imgs:
        pl = engine.plane(w, h, bd)
        pl.upload_pel(img, (80, 80))
        pls.append(pl)
    return pls


@pytest.mark.parametrize("bd,w,h,sr,k", [(8, 320, 192, 24, 3), (8, 1280, 720, 64, 4), (10, 256, 192, 16, 5), (8, 192, 128, 100, 2)])
def test_pairs_per_launch_equals_single_launches_and_oracle(engine, oracle_lib, bd, w, h, sr, k):
    """hmme_search_pairs_device / hmme_refine_pairs_device: k DIFFERENT (current, reference) pairs in one launch == k single
    launches == the oracle (all CTUs at the small sizes, a CTU sample at 720p where the launch also has a tail plan)"""
    import torch
    from hmme import api, synth
    engine.set_lambda(57.9)
    lq = engine.lambda_q16
    n_ctu = ((w + 63) // 64) * ((h + 63) // 64)
    imgs = [synth.make_pair(w, h, seed=100 + i, bit_depth=bd, max_mv=min(sr, 9), region=64) for i in range(k)]
    curs = _planes(engine, w, h, bd, [p[0] for p in imgs])
    refs = _planes(engine, w, h, bd, [p[1] for p in imgs])
    pred = np.stack([synth.random_predictors(n_ctu, seed=7 + i, max_pel=min(sr, 12)) for i in range(k)])
    dev = torch.device("cuda", 0)
    d_pred = torch.from_numpy(pred).to(dev)
    d_mv = torch.zeros((k, n_ctu, 593, 2), dtype=torch.int16, device=dev)
    d_sad = torch.zeros((k, n_ctu, 593), dtype=torch.int32, device=dev)
    d_q = torch.zeros_like(d_mv)
    d_c = torch.zeros_like(d_sad)
    fp = api.FrameParams(sr, 1, bd, 0, n_ctu)
    s = torch.cuda.current_stream().cuda_stream
    engine.search_pairs_device(curs, refs, fp, d_pred.data_ptr(), d_mv.data_ptr(), d_sad.data_ptr(), s)
    engine.refine_pairs_device(curs, refs, fp, d_pred.data_ptr(), d_mv.data_ptr(), 1, d_q.data_ptr(), d_c.data_ptr(), s)
    torch.cuda.synchronize()
    mv, sad = d_mv.cpu().numpy(), d_sad.cpu().numpy().view(np.uint32)
    qmv, cost = d_q.cpu().numpy(), d_c.cpu().numpy().view(np.uint32)
    sample = list(range(n_ctu)) if n_ctu <= 20 else [0, n_ctu // 2 + 3, n_ctu - 1]
    for i in range(k):
        mv1, sad1 = engine.search_frame(curs[i], refs[i], sr, pred[i])
        assert np.array_equal(mv[i], mv1) and np.array_equal(sad[i], sad1), f"pair {i} of the batched launch differs from its own launch"
        q1, c1 = engine.refine_frame(curs[i], refs[i], sr, mv1, pred[i])
        assert np.array_equal(qmv[i], q1) and np.array_equal(cost[i], c1), f"pair {i}: batched refinement differs"
        for ctu in sample:
            ox, oy, osad = oracle_lib.search_frame(imgs[i][0], imgs[i][1], (80, 80), w, h, sr, pred[i], lq, 1, bd, ctu, 1, 1)
            assert np.array_equal(mv[i, ctu, :, 0], ox[0]) and np.array_equal(mv[i, ctu, :, 1], oy[0]) and np.array_equal(sad[i, ctu], osad[0]), (i, ctu)
    assert len({tuple(mv[i, 1, 592]) for i in range(k)}) > 1      # the pairs really differ
    # argument checks: planes of another size / too many pairs
    other = engine.plane(w + 64, h, bd)
    with pytest.raises(api.HmmeError):
        engine.search_pairs_device([curs[0], other], [refs[0], other], fp, None, d_mv.data_ptr(), d_sad.data_ptr(), s)
    with pytest.raises(api.HmmeError):
        engine.search_pairs_device([curs[0]] * 17, [refs[0]] * 17, fp, None, d_mv.data_ptr(), d_sad.data_ptr(), s)
    other.close()
    for pl in curs + refs:
        pl.close()


def _fuzz_pairs_case(engine, oracle_lib, seed):
    import torch
    from hmme import api, synth
    rng = np.random.default_rng(seed)
    w, h = 8 * int(rng.integers(1, 30)), 8 * int(rng.integers(1, 20))
    bd = int(rng.choice([8, 8, 10, 12]))
    k = int(rng.integers(1, 7))
    sr = int(rng.choice([1, 3, 8, 16, 31, 64, 65, 100]))
    n = api.load().hmme_num_ctus(w, h)
    if k * n * (2 * sr + 1) ** 2 > 8 * 129 * 129:     # keep the oracle's share of the run in seconds
        sr = 12
    fen = int(rng.integers(0, 2))
    max_pel = int(rng.choice([0, 6, 60]))
    lam = float(rng.choice([0.0, 57.9, 900.0]))
    engine.set_lambda(lam)
    m = synth.MARGIN
    # a small pool of pictures; pairs draw current and reference from it, so planes are shared between pairs in both roles
    pool = [synth.make_pair(w, h, seed=seed * 10 + i, bit_depth=bd, max_mv=min(sr, 8), region=32, noise_sigma=float(rng.choice([0.0, 2.0])))
            for i in range(max(2, (k + 1) // 2))]
    imgs = [p[0] for p in pool] + [p[1] for p in pool]
    planes = _planes(engine, w, h, bd, imgs)
    pick = [(int(rng.integers(0, len(imgs))), int(rng.integers(0, len(imgs)))) for _ in range(k)]
    pred = np.stack([synth.random_predictors(n, seed=seed + i, max_pel=max(max_pel, 1)) for i in range(k)]) if max_pel else None
    dev = torch.device("cuda", 0)
    d_pred = torch.from_numpy(pred).to(dev) if pred is not None else None
    d_mv = torch.zeros((k, n, 593, 2), dtype=torch.int16, device=dev)
    d_sad = torch.zeros((k, n, 593), dtype=torch.int32, device=dev)
    fp = api.FrameParams(sr, fen, bd, 0, n)
    engine.search_pairs_device([planes[c] for c, _ in pick], [planes[r] for _, r in pick], fp, d_pred.data_ptr() if pred is not None else None,
                               d_mv.data_ptr(), d_sad.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    mv, sad = d_mv.cpu().numpy(), d_sad.cpu().numpy().view(np.uint32)
    tag = dict(seed=seed, w=w, h=h, bd=bd, k=k, sr=sr, fen=fen, max_pel=max_pel, lam=lam, pick=pick)
    for i, (c, r) in enumerate(pick):
        ox, oy, osad = oracle_lib.search_frame(imgs[c], imgs[r], (m, m), w, h, sr, pred[i] if pred is not None else None, engine.lambda_q16, fen, bd,
                                               n_threads=8)
        assert np.array_equal(mv[i, :, :, 0], ox) and np.array_equal(mv[i, :, :, 1], oy) and np.array_equal(sad[i], osad), (tag, i)
    for pl in planes:
        pl.close()


def test_fuzz_pair_launches_vs_oracle(engine, oracle_lib):
    """random numbers of picture pairs (1..6) drawn from a shared pool of planes, random picture sizes, bit depths, search ranges
    (tiled 8-bit windows included), FEN, predictors: every CTU of every pair of the launch against the oracle"""
    n = int(os.environ.get("HMME_FUZZ_CASES", "10"))
    base = int(os.environ.get("HMME_FUZZ_SEED", "3000"))
    for i in range(n):
        _fuzz_pairs_case(engine, oracle_lib, base + i)


def test_plane_refill_on_another_stream_waits_for_the_search_that_reads_it(engine):
    """write-after-read across streams (include/hmme.h "Streams"): a 2160p search (~2.5 ms) is enqueued on stream A, then the
    plane it reads is refilled from page-locked memory on stream B (hmme_plane_upload_async, ~1 ms) without any host-side wait.
    The first search must see the old picture in every CTU, a second one (stream A again) the new picture."""
    import torch
    from hmme import api, synth
    w, h, sr = 3840, 2160, 64
    engine.set_lambda(57.9)
    seq = synth.Sequence(w, h, 3, seed=31)
    a_img, b_img, c_img = seq.luma(0), seq.luma(1), seq.luma(2)
    cur, ref = engine.plane(w, h), engine.plane(w, h)
    cur.upload_u8(a_img)
    ref.upload_u8(b_img)
    want_b = engine.search_frame(cur, ref, sr)              # reference picture = b
    ref.upload_u8(c_img)
    want_c = engine.search_frame(cur, ref, sr)              # reference picture = c
    assert not np.array_equal(want_b[0], want_c[0])
    ref.upload_u8(b_img)
    dev = torch.device("cuda", 0)
    n_ctu = 2040
    fp = api.FrameParams(sr, 1, 8, 0, n_ctu)
    host_c = torch.from_numpy(c_img).pin_memory()
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [(torch.zeros((n_ctu, 593, 2), dtype=torch.int16, device=dev), torch.zeros((n_ctu, 593), dtype=torch.int32, device=dev)) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(3):                                     # a few rounds: b -> c -> b ...
        first, second = (want_b, want_c) if rep % 2 == 0 else (want_c, want_b)
        nxt = host_c if rep % 2 == 0 else torch.from_numpy(b_img).pin_memory()
        engine.search_frame_device(cur, ref, fp, None, outs[0][0].data_ptr(), outs[0][1].data_ptr(), s_a.cuda_stream)
        ref.upload_async(nxt.data_ptr(), w, 1, s_b.cuda_stream)          # no synchronisation in between
        engine.search_frame_device(cur, ref, fp, None, outs[1][0].data_ptr(), outs[1][1].data_ptr(), s_a.cuda_stream)
        torch.cuda.synchronize()
        engine.upload_status(s_b.cuda_stream)
        assert np.array_equal(outs[0][0].cpu().numpy(), first[0]) and np.array_equal(outs[0][1].cpu().numpy().view(np.uint32), first[1]), \
            f"round {rep}: the search saw a plane that was being refilled on another stream"
        assert np.array_equal(outs[1][0].cpu().numpy(), second[0]) and np.array_equal(outs[1][1].cpu().numpy().view(np.uint32), second[1]), \
            f"round {rep}: the search after the refill did not see the new picture"
    cur.close(); ref.close()


def test_async_upload_reports_out_of_range_samples_once(engine):
    import torch
    from hmme import api
    w, h = 128, 64
    pl = engine.plane(w, h, 10)
    good = torch.full((h, w), 1023, dtype=torch.int16).pin_memory()
    bad = good.clone().pin_memory()
    bad[5, 7] = 1024
    s = torch.cuda.Stream()
    pl.upload_async(good.data_ptr(), w, 2, s.cuda_stream)
    engine.upload_status(s.cuda_stream)
    pl.upload_async(bad.data_ptr(), w, 2, s.cuda_stream)
    with pytest.raises(api.HmmeError, match="range"):
        engine.upload_status(s.cuda_stream)
    engine.upload_status(s.cuda_stream)                      # latched once, then clear
    # the latch of the asynchronous uploads is their own: a bad asynchronous upload does not fail the next SYNCHRONOUS upload of
    # another, valid plane (round 3 shared one flag), and that synchronous upload does not swallow the violation either
    other = engine.plane(w, h, 10)
    pl.upload_async(bad.data_ptr(), w, 2, s.cuda_stream)
    s.synchronize()
    other.upload_pel(good.numpy(), (0, 0))
    with pytest.raises(api.HmmeError, match="range"):
        engine.upload_status(s.cuda_stream)
    with pytest.raises(api.HmmeError, match="outside"):     # and a bad synchronous upload reports at once, leaving the other latch alone
        other.upload_pel(bad.numpy(), (0, 0))
    engine.upload_status(s.cuda_stream)
    other.close()
    with pytest.raises(api.HmmeError):
        pl.upload_async(good.data_ptr(), w, 3, s.cuda_stream)
    pl.close()


def test_searches_on_device_addresses_with_bit_31_set(oracle_lib):
    """Pins the round-2 fault (gpurun_out/r02H): the kernels read the current block through scalar loads from a 64-bit base
    assembled out of two readfirstlane halves; widening the `int` the builtin returns sign-extended a low half with bit 31 set
    and the load went to 0xffffffff........  Whether a launch meets such an address depends on where hipMalloc puts a plane, so
    this test walks the allocator (filler planes of 186 MB, straight hipMalloc) until planes AND a context's per-CTU staging block sit on addresses
    whose low dword has bit 31 set -- asserted -- and then runs the frame path and the per-CTU call, 8- and 10-bit, against the
    oracle on exactly those buffers."""
    from hmme import api, synth
    w, h, sr = 256, 192, 16
    m = synth.MARGIN
    lq = oracle_lib.oracle().hmo_lambda_q16(57.9)
    fillers, tried = [], []
    found = {}
    filler_eng = api.Engine(0, 64)

    def high(addr, span):       # every byte the kernels address from this base has bit 31 set
        return (addr & 0x80000000) and ((addr + span) & 0x80000000) and ((addr & 0xffffffff) + span < (1 << 32))

    graveyard = []              # nothing is freed while walking: a freed block's address would simply be handed out again
    for step in range(200):                                  # up to 37 GB of address space: bit 31 flips every 2 GiB
        if len(found) == 3:
            break
        eng = api.Engine(0, 64)
        p8, p10 = eng.plane(w, h, 8), eng.plane(w, h, 10)
        graveyard.append((eng, p8, p10))
        a = {"ctx": eng.call_block_address, "p8": p8.device_address, "p10": p10.device_address}
        tried.append({k: hex(v) for k, v in a.items()})
        if "ctx" not in found and high(a["ctx"], 8192):
            found["ctx"] = eng
        if "p8" not in found and high(a["p8"], 512 * 300):
            found["p8"] = (eng, p8)
        if "p10" not in found and high(a["p10"], 1024 * 300):
            found["p10"] = (eng, p10)
        fillers.append(filler_eng.plane(16384, 11000, 8))      # 16 640 B x 11 161 rows, not touched
    print(f"high-address walk: {len(tried)} rounds; last addresses {tried[-1]}")
    try:
        assert len(found) == 3, f"no buffer with bit 31 set in its low address dword after {len(tried)} rounds: {tried[-6:]}"
        for bd, key in ((8, "p8"), (10, "p10")):
            eng, cur_plane = found[key]
            assert cur_plane.device_address & 0x80000000
            eng.set_lambda(57.9)
            cur, ref, _ = synth.make_pair(w, h, seed=40 + bd, bit_depth=bd, max_mv=8, region=64)
            cur_plane.upload_pel(cur, (m, m))
            ref_plane = eng.plane(w, h, bd)
            ref_plane.upload_pel(ref, (m, m))
            mv, sad = eng.search_frame(cur_plane, ref_plane, sr, None, ctu_first=5, ctu_count=4)
            ox, oy, osad = oracle_lib.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, 5, 4, 4)
            assert np.array_equal(mv[:, :, 0], ox) and np.array_equal(mv[:, :, 1], oy) and np.array_equal(sad, osad), f"{bd}-bit frame path"
            q, c = eng.refine_frame(cur_plane, ref_plane, sr, mv, None, ctu_first=5, ctu_count=4)
            oq, oc = oracle_lib.refine_frame(cur, ref, (m, m), w, h, mv, None, lq, 1, bd, 5, 4, 4)
            assert np.array_equal(q, oq) and np.array_equal(c, oc), f"{bd}-bit refinement"
            ref_plane.close()
        eng = found["ctx"]
        assert eng.call_block_address & 0x80000000
        eng.set_lambda(57.9)
        for bd in (8, 10):
            cur, ref, _ = synth.make_pair(w, h, seed=50 + bd, bit_depth=bd, max_mv=8, region=64)
            p = api.SearchParams(-sr, -sr, sr, sr, 3, -5, 1, bd)
            mv, sad = eng.search_ctu(cur, (m + 64, m + 64), ref, (m + 64, m + 64), p)
            op = oracle_lib.make_params((-sr, -sr), (sr, sr), (3, -5), lq, 1, bd)
            ox, oy, osad = oracle_lib.search_ctu(cur, (m + 64, m + 64), ref, (m + 64, m + 64), op)
            assert np.array_equal(mv[:, 0], ox) and np.array_equal(mv[:, 1], oy) and np.array_equal(sad, osad), f"{bd}-bit per-CTU call"
    finally:
        for eng, p8, p10 in graveyard:
            p8.close(); p10.close(); eng.close()
        for pl in fillers:
            pl.close()
        filler_eng.close()
