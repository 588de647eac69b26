"""N > 1 path on CPU: world_size-2 gloo run of the frame-shard driver (hmme/shard.py).  The GPU
engine is replaced by the CPU oracle as the per-pair search so the gather/re-interleave logic
is checked end to end against a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


W, H, SR, N_PAIRS = 128, 72, 4, 5


def _pair_search(oracle_py, synth):
    lq = oracle_py.oracle().hmo_lambda_q16(57.9)

    def run(p, out_mv, out_sad):
        cur, ref, _ = synth.make_pair(W, H, seed=100 + p, max_mv=3, region=64)
        ox, oy, osad = oracle_py.search_frame(cur, ref, (80, 80), W, H, SR, None, lq, 1, 8)
        out_mv[:, :, 0] = torch.from_numpy(ox.astype(np.int16))
        out_mv[:, :, 1] = torch.from_numpy(oy.astype(np.int16))
        out_sad.copy_(torch.from_numpy(osad.astype(np.int32)))
    return run


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "hm-opencl_amd")):
        sys.path.insert(0, p)
    import oracle_py
    from hmme import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mv, sad = shard.search_sequence(_pair_search(oracle_py, synth), N_PAIRS, 2 * 2, torch.device("cpu"))
        if rank == 0:
            q.put((mv.numpy().copy(), sad.numpy().copy()))
        else:   # the tables go to rank 0 only: nothing is sent to a rank that does not read them
            assert mv is None and sad is None
        dist.barrier()
    except Exception as e:  # surface the failure instead of letting the parent time out
        if rank == 0:
            q.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()


def test_pairs_round_robin():
    from hmme import shard
    assert shard.pairs_for_rank(5, 0, 2) == [0, 2, 4] and shard.pairs_for_rank(5, 1, 2) == [1, 3]
    assert shard.pairs_per_rank(5, 2) == 3 and shard.pairs_per_rank(64, 8) == 8
    assert shard.pair_counts(5, 2) == [3, 2] and shard.pair_counts(124, 8) == [16] * 4 + [15] * 4 and shard.pair_counts(3, 4) == [1, 1, 1, 0]
    assert sorted(sum((shard.pairs_for_rank(64, r, 8) for r in range(8)), [])) == list(range(64))


def test_gop_pairs_follow_the_reference_cfgs():
    from hmme import shard
    ra = shard.gop_pairs(9, "randomaccess")   # cfg/encoder_randomaccess_main.cfg:28-31
    assert ra[:8] == [(4, 0), (2, 0), (2, 4), (1, 0), (1, 2), (1, 4), (3, 2), (3, 4)]
    assert (8, 4) in ra and all(0 <= c < 9 and 0 <= r < 9 for c, r in ra)
    ld = shard.gop_pairs(6, "lowdelay_P")      # cfg/encoder_lowdelay_P_main.cfg:24-27
    assert ld == [(1, 0), (2, 1), (2, 0), (3, 2), (3, 0), (4, 3), (4, 0), (5, 4), (5, 0)]
    assert len(shard.gop_pairs(64, "randomaccess")) == 124   # BASELINE config 4: one 64-picture sequence


def test_two_rank_gloo_matches_single_process(oracle_lib):
    from hmme import shard, synth
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    assert not isinstance(got, str), got
    mv2, sad2 = got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mv1, sad1 = shard.search_sequence(_pair_search(oracle_lib, synth), N_PAIRS, 4, torch.device("cpu"))
    assert mv2.shape == (N_PAIRS, 4, 593, 2)
    assert np.array_equal(mv2, mv1.numpy()) and np.array_equal(sad2, sad1.numpy())


# ---- the pipeline bench.py --gpus N actually runs: PipelinedGather (double-buffered step / drain) -------------------------
N_STEPS, SHAPE = 9, (2, 1, 6, 593)


def _expected_block(rank, step):
    """what rank `rank` contributes at step `step`: a deterministic pattern over the whole [2, k, n_ctu, 593] block"""
    g = torch.Generator().manual_seed(1000 * rank + step)
    return torch.randint(-2**31, 2**31 - 1, SHAPE, dtype=torch.int32, generator=g)


def _pipeline_worker(rank, world, port, q):
    import time
    sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
    from hmme import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = shard.PipelinedGather(lambda: torch.zeros(SHAPE, dtype=torch.int32), distributed=True, async_op=True)
        seen = {}

        def launch(buf, k):
            # a search kernel fills its output over time, not at once: write the block in slices with pauses, so that a gather
            # still reading this buffer (the bug the retire-before-reuse rule prevents) would ship a torn block
            want = _expected_block(rank, k)
            for i in range(SHAPE[2]):
                buf[:, :, i] = want[:, :, i]
                if (k + rank + i) % 3 == 0:
                    time.sleep(0.002)

        def consume(k, gathered):
            assert (gathered is None) == (rank != 0)      # only rank 0 receives
            seen[k] = gathered.clone() if gathered is not None else None

        for _ in range(N_STEPS):
            pipe.step(launch, consume)
            assert sum(p is not None for p in pipe.pending) <= 2
        pipe.drain(consume)
        assert pipe.pending == [None, None]
        ok = sorted(seen) == list(range(N_STEPS))
        for k, g in seen.items():
            if rank != 0:
                continue
            ok = ok and tuple(g.shape) == (world,) + SHAPE
            for r in range(world):
                ok = ok and bool(torch.equal(g[r], _expected_block(r, k)))
        if rank == 0:   # what rank 1 sent: one block per step, nothing more (round 3's all-gather moved world x that, to every rank)
            ok = ok and pipe.bytes_received == N_STEPS * 4 * int(np.prod(SHAPE))
        else:
            ok = ok and pipe.bytes_received == 0
        q.put((rank, ok, sorted(seen)))
        dist.barrier()
    except Exception as e:
        q.put((rank, repr(e), []))
        raise
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_two_ranks_gloo():
    """the double-buffered step()/drain() logic of bench.py (hmme/shard.py PipelinedGather) under gloo, world size 2, async
    transfers: every step's gathered block arrives complete and in order on rank 0 -- and only there -- although each local buffer
    is rewritten two steps later"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, steps in got:
        assert ok is True, (rank, ok)
        assert steps == list(range(N_STEPS))


def test_pipelined_gather_single_process_passthrough():
    """world size 1 without a process group: consume sees the local block itself, one step late, then at drain"""
    from hmme import shard
    pipe = shard.PipelinedGather(lambda: torch.zeros((2, 1, 2, 593), dtype=torch.int32), distributed=False)
    seen = []
    for k in range(5):
        pipe.step(lambda buf, kk: buf.fill_(kk + 1), lambda kk, g: seen.append((kk, int(g[0, 0, 0, 0, 0]))))
    pipe.drain(lambda kk, g: seen.append((kk, int(g[0, 0, 0, 0, 0]))))
    assert seen == [(k, k + 1) for k in range(5)]
    assert int(pipe.last_local[0, 0, 0, 0]) == 5


# ---- gather_to_root with unequal and empty shares: 2 pairs on 3 ranks (rank 2 has nothing to send) -----------------------------
def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
    from hmme import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_pairs = 2
        mine = shard.pairs_for_rank(n_pairs, rank, world)
        mv = torch.zeros((len(mine), 3, 593, 2), dtype=torch.int16)
        sad = torch.zeros((len(mine), 3, 593), dtype=torch.int32)
        for i, p in enumerate(mine):
            mv[i] = 100 * p + 7
            mv[i, :, :, 1] = -(100 * p + 7)
            sad[i] = 1000 + p
        g_mv, g_sad = shard.gather_pair_results(mv, sad, n_pairs)
        if rank == 0:
            ok = tuple(g_mv.shape) == (2, 3, 593, 2) and tuple(g_sad.shape) == (2, 3, 593)
            for p in range(n_pairs):
                ok = ok and bool((g_mv[p, :, :, 0] == 100 * p + 7).all()) and bool((g_mv[p, :, :, 1] == -(100 * p + 7)).all()) and bool((g_sad[p] == 1000 + p).all())
            _, _, got = shard.gather_to_root(sad, shard.pair_counts(n_pairs, world))
            q.put((ok, got))
        else:
            assert g_mv is None
            shard.gather_to_root(sad, shard.pair_counts(n_pairs, world))
        dist.barrier()
    except Exception as e:
        if rank == 0:
            q.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()


def test_gather_to_root_with_unequal_and_empty_shares_three_ranks_gloo():
    """124 pairs on 8 ranks are 16 / 15 per rank; the extreme of that: 2 pairs on 3 ranks (1, 1, 0).  Rank 0 receives exactly the
    tables that exist -- no zero tables pad a short rank, nothing is sent to ranks 1 and 2"""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == (True, 1 * 3 * 593 * 4), got     # one table from rank 1, none from rank 2


# ---- start-up of an unattended N-rank run: a rank that hangs before its first barrier must end the job loudly (bench.py --gpus N) ------
def _startup_worker(rank, world, port, stall_rank, stall_s, limit_s, q):
    sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
    from hmme import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import io
    import time
    msgs = io.StringIO()

    def give_up(code):   # what os._exit would do, made visible to the parent first
        q.put((rank, "watchdog", msgs.getvalue()))
        q.close()
        q.join_thread()   # the queue's feeder thread must have written the message before the process is gone
        os._exit(code)

    wd = shard.StartupWatchdog(rank, world, limit_s, label="test", _exit=give_up, _out=msgs)
    wd.stage("init_process_group (rendezvous)")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wd.stage("rendezvous store count")
    seen = shard.rendezvous_report(rank, world, timeout_s=20.0, out=msgs)
    if rank == stall_rank:
        time.sleep(stall_s)
    wd.stage("first barrier")
    dist.barrier()
    wd.done()
    q.put((rank, "passed", seen))
    dist.destroy_process_group()


def _run_startup(stall_rank, stall_s, limit_s, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_startup_worker, args=(r, world, port, stall_rank, stall_s, limit_s, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    import queue
    try:
        while len(got) < world:
            got.append(q.get(timeout=60))
            if got[-1][1] == "watchdog":
                break
    except queue.Empty:
        pass
    for p in procs:
        p.join(timeout=20)
        if p.is_alive():      # the rank left behind in the barrier by the one that gave up: what the launcher would take down
            p.kill()
            p.join()
    return got, [p.exitcode for p in procs]


def test_startup_watchdog_lets_a_healthy_start_through():
    got, codes = _run_startup(stall_rank=-1, stall_s=0.0, limit_s=30.0)
    assert sorted(g[0] for g in got) == [0, 1] and all(g[1] == "passed" for g in got), got
    assert [g[2] for g in got if g[0] == 0] == [2]            # rank 0 counted both ranks through the store before the first collective
    assert codes == [0, 0]


def test_startup_watchdog_ends_the_rank_that_waits_for_a_sleeping_one():
    """rank 1 sits still in front of the first barrier: rank 0, stuck in that barrier, gives up after its limit with exit code 3 and a line
    that names itself and the stage -- the job fails instead of hanging"""
    got, codes = _run_startup(stall_rank=1, stall_s=30.0, limit_s=3.0)
    assert got and got[-1][1] == "watchdog", got
    rank, _, text = got[-1]
    assert f"rank {rank} of 2 did not get past" in text and ("first barrier" in text or "rendezvous" in text), text
    assert 3 in codes, codes


def test_shard_docstring_names_the_exchange_that_is_used():
    from hmme import shard
    assert "gather_to_root" in shard.__doc__ and "all_gather_into_tensor" not in shard.__doc__


def test_bench_parent_takes_its_ranks_down_when_it_is_terminated(tmp_path):
    """`python bench.py --gpus N` started without a launcher runs the ranks in a session of their own; a SIGTERM that ends the parent (a
    harness time-out) must end that session too and remove the status directory -- nothing stays behind on the GPUs.  The launcher is a
    stand-in here (HMME_BENCH_TEST_LAUNCHER): a process that starts a grandchild, records both pids and the status directory, and sleeps."""
    import json
    import signal
    import subprocess
    import time
    note = tmp_path / "pids.json"
    standin = ("import json, os, subprocess, sys, time\n"
               "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)'])\n"
               f"json.dump({{'child': os.getpid(), 'grandchild': g.pid, 'status': os.environ['HMME_BENCH_STATUS_DIR']}}, open({str(note)!r}, 'w'))\n"
               "time.sleep(300)\n")
    env = dict(os.environ, HMME_BENCH_TEST_LAUNCHER=json.dumps([sys.executable, "-c", standin]))
    parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        t0 = time.time()
        while not (note.exists() and note.stat().st_size > 0) and time.time() - t0 < 60:
            time.sleep(0.1)
        time.sleep(0.2)
        info = json.loads(note.read_text())
        assert os.path.isdir(info["status"])
        parent.send_signal(signal.SIGTERM)
        out, err = parent.communicate(timeout=60)
    finally:
        if parent.poll() is None:
            parent.kill()
    assert parent.returncode == 128 + signal.SIGTERM, (parent.returncode, err[-500:])
    assert "ending the child process group" in err and out == ""

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:                                   # a zombie of another parent still answers kill(0): look at its state
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False
    t0 = time.time()
    while (alive(info["child"]) or alive(info["grandchild"])) and time.time() - t0 < 15:
        time.sleep(0.1)
    assert not alive(info["child"]) and not alive(info["grandchild"]), info
    assert not os.path.exists(info["status"])


# ---- BASELINE config 4 as an N-rank job (bench.py --gpus N `configs.config4_sharded`): 124 pairs dealt p mod N, gathered in pair order -------
def _job_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
    from hmme import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = shard.gop_pairs(64, "randomaccess")
        index = {pr: i for i, pr in enumerate(pairs)}
        calls = []

        def run_share(share):       # the engine's stand-in: the tables of pair p are a function of p and of the pair's two pictures
            calls.append(len(share))
            mv = torch.zeros((len(share), 2, 593, 2), dtype=torch.int16)
            sad = torch.zeros((len(share), 2, 593), dtype=torch.int32)
            for i, (c, r) in enumerate(share):
                mv[i, :, :, 0] = index[(c, r)]
                mv[i, :, :, 1] = c - r
                sad[i] = 100000 * c + r
            return {"mv": mv, "sad": sad, "stages": {"search_s": 0.0}}
        job = shard.sharded_sequence_job(run_share, pairs, passes=2)
        assert calls == [len(shard.pairs_for_rank(len(pairs), rank, world))] * 3      # one untimed pass, two timed ones, always this rank's share
        if rank == 0:
            mv, sad = job["mv"], job["sad"]
            ok = tuple(mv.shape) == (124, 2, 593, 2) and tuple(sad.shape) == (124, 2, 593)
            for p, (c, r) in enumerate(pairs):
                ok = ok and bool((mv[p, :, :, 0] == p).all()) and bool((mv[p, :, :, 1] == c - r).all()) and bool((sad[p] == 100000 * c + r).all())
            q.put((ok, job["pair_counts"], job["crc32_tables_match_per_rank"], [e["pairs"] for e in job["per_rank"]], len(job["seconds_passes"]),
                   job["seconds"] in job["seconds_passes"]))
        else:
            assert job is None
        dist.barrier()
    except Exception as e:
        if rank == 0:
            q.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()



@pytest.mark.parametrize("world,counts", [(2, [62, 62]), (3, [42, 41, 41]), (8, [16, 16, 16, 16, 15, 15, 15, 15])])
def test_config4_sharded_job_deals_124_pairs_and_gathers_them_in_pair_order(world, counts):
    """the random-access GOP of 64 pictures is 124 pairs (cfg/encoder_randomaccess_main.cfg:28-31): dealt p mod N they are ragged on 3 ranks
    and on the 8 of BASELINE config 4 (16 / 15); rank 0 ends up with all 124 tables in pair order, every rank's CRC matches what rank 0 holds"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0
    assert got == (True, counts, [True] * world, counts, 2, True), got


def test_sharded_job_without_a_process_group_is_the_one_rank_job():
    from hmme import shard
    pairs = shard.gop_pairs(8, "randomaccess")
    run = lambda share: {"mv": torch.full((len(share), 1, 593, 2), 3, dtype=torch.int16), "sad": torch.ones((len(share), 1, 593), dtype=torch.int32)}
    job = shard.sharded_sequence_job(run, pairs, passes=1)
    assert job["pair_counts"] == [len(pairs)] and job["crc32_tables_match_per_rank"] == [True] and tuple(job["mv"].shape) == (len(pairs), 1, 593, 2)
