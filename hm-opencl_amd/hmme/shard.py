"""Frame sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the MI355X node, "gloo" in the CPU tests).

Open-loop motion estimation over a sequence is embarrassingly parallel per (current, reference)
picture pair: pair p is searched by rank p % world with no exchange during the
search.  The only exchange step is the result gather TO RANK 0 (gather_to_root): every other rank
sends its [k_r, n_ctu, 593] block of (mv, sad) -- 9.7 MB per 2160p pair, k_r pairs, ragged and
possibly zero -- as grouped point-to-point sends (dist.batch_isend_irecv = ncclGroupStart / ncclSend /
ncclRecv / ncclGroupEnd on RCCL); rank 0 re-interleaves the blocks into pair order.  Nothing is sent
to ranks that do not read it.  PipelinedGather overlaps step k's transfer with step k + 1's search.

Start-up of an unattended N-rank run (bench.py --gpus N): `StartupWatchdog` ends a rank that does not
get through rendezvous and its first collective in time, naming itself and the stage it hung in, and
`rendezvous_report` counts the ranks through the job's store BEFORE the first collective.
"""
import os
import sys
import threading
import time

import torch
import torch.distributed as dist


class StartupWatchdog:
    """A rank that hangs before its first barrier (a wedged device, a rank that never joins, an RCCL communicator that cannot be built)
    would hang the whole job silently.  The watchdog thread ends THIS process with exit code 3 once `seconds` have passed without
    `done()` -- after one stderr line that names the rank and the stage it was in; the launcher (torch.distributed.run, or bench.py's own
    parent process) then takes the other ranks down and the job fails loudly instead of sitting in a collective.  Nothing is re-executed
    and no other process is signalled from here."""

    def __init__(self, rank, world, seconds, label="bench.py", _exit=os._exit, _out=None):
        self.rank, self.world, self.seconds, self.label = rank, world, float(seconds), label
        self._stage = "start"
        self._done = threading.Event()
        self._exit, self._out = _exit, _out or sys.stderr
        self._t0 = time.monotonic()
        self._thread = threading.Thread(target=self._run, name="hmme-startup-watchdog", daemon=True)
        if self.seconds > 0:
            self._thread.start()

    def stage(self, name):
        self._stage = name

    def done(self):
        self._done.set()

    def _run(self):
        if self._done.wait(self.seconds):
            return
        self._out.write(f"{self.label}: rank {self.rank} of {self.world} did not get past '{self._stage}' within {self.seconds:.0f} s "
                        f"(pid {os.getpid()}, LOCAL_RANK {os.environ.get('LOCAL_RANK', '?')}): giving up so that the job fails instead of hanging\n")
        self._out.flush()
        self._exit(3)


def rendezvous_report(rank, world, timeout_s=120.0, key="hmme/ranks_seen", out=None):
    """Counts the ranks through the job's key-value store (TCP: no collective, no GPU) -- every rank adds itself, rank 0 waits until all
    `world` have, or says which count it got stuck at.  Returns the number of ranks seen (rank 0) or None (other ranks; also rank 0 when
    this torch does not hand out the job's store -- the lookup is a private function -- in which case the count is skipped with a note and
    the first barrier, under the watchdog, is the check)."""
    out = out or sys.stderr
    try:
        store = dist.distributed_c10d._get_default_store()
    except (AttributeError, RuntimeError) as e:
        if rank == 0:
            out.write(f"hmme: the rendezvous store is not reachable through this torch ({e.__class__.__name__}: {e}): ranks are not counted before the first collective\n")
            out.flush()
        return None
    store.add(key, 1)
    if rank != 0:
        return None
    t0 = time.monotonic()
    seen = 0
    while True:
        seen = int(store.add(key, 0))
        if seen >= world or time.monotonic() - t0 > timeout_s:
            break
        time.sleep(0.05)
    out.write(f"hmme: ranks_seen {seen} of {world} through the rendezvous store, before the first collective\n")
    out.flush()
    return seen


# reference-picture offsets (POC deltas) of the reference's two GOP presets, per position in the GOP of 4
GOP_REFS = {
    # cfg/encoder_randomaccess_main.cfg:28-31  (Frame1..4: POC 4, 2, 1, 3)
    "randomaccess": {4: (-4,), 2: (-2, 2), 1: (-1, 1, 3), 3: (-1, 1)},
    # cfg/encoder_lowdelay_P_main.cfg:24-27    (4 active references per P picture)
    "lowdelay_P": {1: (-1, -5, -9, -13), 2: (-1, -2, -6, -10), 3: (-1, -3, -7, -11), 4: (-1, -4, -8, -12)},
}


def gop_pairs(n_frames, structure="randomaccess"):
    """(current POC, reference POC) pairs an open-loop ME pass over `n_frames` pictures searches,
    in coding order of the reference's GOP structure; references outside the sequence are dropped."""
    refs = GOP_REFS[structure]
    pairs = []
    for base in range(0, n_frames, 4):
        for pos in (refs.keys() if structure != "randomaccess" else (4, 2, 1, 3)):
            cur = base + pos
            if cur >= n_frames:
                continue
            for d in refs[pos]:
                if 0 <= cur + d < n_frames:
                    pairs.append((cur, cur + d))
    return pairs


def pairs_for_rank(n_pairs, rank, world):
    """indices of the picture pairs rank `rank` searches (round-robin: pair p -> rank p % world)"""
    return list(range(rank, n_pairs, world))


def pairs_per_rank(n_pairs, world):
    """the largest number of pairs any rank searches (rank r searches len(pairs_for_rank(n_pairs, r, world)) of them)"""
    return (n_pairs + world - 1) // world


def pair_counts(n_pairs, world):
    """pairs per rank, rank by rank -- the load balance of the job (124 pairs on 8 ranks: four ranks search 16, four 15)"""
    return [len(range(r, n_pairs, world)) for r in range(world)]


def _staged(t):
    """gloo moves host tensors: a device tensor of the one-GPU rehearsal (bench.py --share-gpu --backend gloo) travels through the host"""
    return t.cpu() if (dist.get_backend() == "gloo" and t.device.type != "cpu") else t


def gather_to_root(local, counts=None, root=0, out=None, async_op=False):
    """The one exchange step of the path: every rank's result block to rank `root` -- and only there; nothing is sent to ranks that
    do not read it (round 3 all-gathered: 8 x the bytes, plus zero tables that padded short ranks).

    local   this rank's [k_r, ...] block (k_r may differ between ranks and may be 0)
    counts  k_r of every rank (default: all equal to local.shape[0])
    out     root only, optional: preallocated [sum(counts), ...] tensor on local's device
    Returns (blocks, works, bytes_received): on root `blocks` is the list of per-rank views into `out` (rank order) and
    bytes_received what the other ranks sent; elsewhere (None, works, 0).  Point-to-point sends and receives issued as one batch
    (dist.batch_isend_irecv = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL); with async_op the caller waits on
    `works` (a stream-side dependency under RCCL) before it reads `blocks` or rewrites `local`."""
    if not dist.is_initialized():
        return [local], [], 0
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = list(counts) if counts is not None else [int(local.shape[0])] * world
    assert len(counts) == world and int(local.shape[0]) == counts[rank], (counts, rank, tuple(local.shape))
    host_staged = dist.get_backend() == "gloo" and local.device.type != "cpu"
    src = _staged(local.contiguous())
    ops, blocks, received, recv_bufs = [], None, 0, []
    if rank == root:
        if out is None:
            out = torch.empty((sum(counts),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        offs = [sum(counts[:r]) for r in range(world)]
        blocks = [out[offs[r]:offs[r] + counts[r]] for r in range(world)]
        blocks[root].copy_(local)
        for r in range(world):
            if r == root or counts[r] == 0:
                continue
            buf = torch.empty(blocks[r].shape, dtype=local.dtype) if host_staged else blocks[r]
            recv_bufs.append((buf, blocks[r]))
            ops.append(dist.P2POp(dist.irecv, buf, r))
            received += buf.numel() * buf.element_size()
    elif counts[rank] > 0:
        ops.append(dist.P2POp(dist.isend, src, root))
    works = dist.batch_isend_irecv(ops) if ops else []
    if not async_op or host_staged:
        for w in works:
            w.wait()
        works = []
        if host_staged:
            for buf, dst in recv_bufs:
                dst.copy_(buf)
    return blocks, works, received


def gather_pair_results(local_mv, local_sad, n_pairs, root=0):
    """local_mv: [k_r, n_ctu, 593, 2] int16, local_sad: [k_r, n_ctu, 593] int32: the tables of this rank's pairs (pairs_for_rank order;
    rows beyond them, if any, are ignored).  Returns (mv, sad) in pair order on rank `root`, (None, None) on the others."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local_mv[:n_pairs], local_sad[:n_pairs]
    rank = dist.get_rank()
    counts = pair_counts(n_pairs, world)
    # an (mvx, mvy) int16 pair travels as one int32 word (TComMv is 4 bytes; gloo has no int16 point-to-point either)
    mv_words = local_mv[:counts[rank]].contiguous().view(torch.int32).squeeze(-1)
    b_mv, _, _ = gather_to_root(mv_words, counts, root)
    b_sad, _, _ = gather_to_root(local_sad[:counts[rank]].contiguous(), counts, root)
    if rank != root:
        return None, None
    mv = torch.empty((n_pairs,) + tuple(mv_words.shape[1:]), dtype=torch.int32, device=local_mv.device)
    sad = torch.empty((n_pairs,) + tuple(local_sad.shape[1:]), dtype=local_sad.dtype, device=local_sad.device)
    for r in range(world):   # block r, row i holds pair i * world + r
        mv[r::world] = b_mv[r]
        sad[r::world] = b_sad[r]
    return mv.unsqueeze(-1).view(torch.int16), sad


class PipelinedGather:
    """Double-buffered result exchange of a stream of search steps (bench.py --gpus N, an open-loop ME pass).

    Step k writes its tables into local buffer k % 2 and starts their gather to rank 0 (gather_to_root); the transfer runs on the
    collective backend's own stream / thread while step k+1 searches into the other buffer, so the 9.7 MB per rank and 2160p pair
    never stall the search kernel.  Before buffer b is written again (step k+2) the transfer that read it (step k) is waited for --
    with RCCL a stream-side dependency, not a host block -- and only then is its gathered block handed to `consume` and recycled.

    make_local() -> tensor            this rank's [2, k, n_ctu, 593] int32 block (plane 0 TComMv words, plane 1 SADs)
    launch(buf, step)                 enqueue the search of step `step` writing into buf (engine call; tests: a stand-in)
    consume(step, gathered) or None   rank 0: gathered = [world, ...] block of `step`, valid only during the call; other ranks: None
    """

    def __init__(self, make_local, distributed, async_op=True, root=0):
        self.distributed = bool(distributed) and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.rank = dist.get_rank() if self.distributed else 0
        self.root = root
        self.async_op = async_op
        self.bufs = [make_local(), make_local()]
        self.gathered = [torch.empty((self.world,) + tuple(b.shape), dtype=b.dtype, device=b.device)
                         if (self.distributed and self.rank == root) else None for b in self.bufs]
        self.pending = [None, None]     # (step, works) of the transfer that last used buffer b
        self.bytes_received = 0         # root: what the other ranks have sent so far
        self.k = 0

    def _retire(self, b, consume):
        if self.pending[b] is None:
            return
        step, works = self.pending[b]
        for w in works:
            w.wait()
        self.pending[b] = None
        if consume is not None:
            consume(step, (self.gathered[b] if self.rank == self.root else None) if self.distributed else self.bufs[b].unsqueeze(0))

    def step(self, launch, consume=None):
        b = self.k & 1
        self._retire(b, consume)        # the transfer that read bufs[b] (two steps ago) is done before bufs[b] is overwritten
        launch(self.bufs[b], self.k)
        works = []
        if self.distributed:   # the one exchange step of the path: this step's tables of all `world` pairs to rank 0
            out = self.gathered[b].view((self.world * self.bufs[b].shape[0],) + tuple(self.bufs[b].shape[1:])) if self.rank == self.root else None
            _, works, got = gather_to_root(self.bufs[b], None, self.root, out, async_op=self.async_op)
            self.bytes_received += got
        self.pending[b] = (self.k, works)
        self.k += 1
        return b

    def drain(self, consume=None):
        for b in ((self.k & 1), (self.k & 1) ^ 1):   # oldest first
            self._retire(b, consume)

    @property
    def last_local(self):
        """this rank's buffer of the most recent step"""
        return self.bufs[(self.k - 1) & 1]

    @property
    def last_gathered(self):
        """rank 0, after drain(): the [world, ...] block of the most recent step (None elsewhere / without a process group)"""
        return self.gathered[(self.k - 1) & 1]


def sharded_sequence_job(run_share, pairs, passes=3, sync=None, reduce_device="cpu"):
    """One open-loop pass over `pairs` as an N-rank job, timed as a job: pair p -> rank p % world, every rank runs ITS share
    (run_share(my_pairs) -> dict with "mv" [k, n_ctu, 593, 2] int16, "sad" [k, n_ctu, 593] int32 and optionally "stages"), the tables are
    gathered to rank 0 in pair order (gather_pair_results).  Each pass: barrier, the rank's share, the gather, barrier; its time is the
    max over ranks.  One untimed pass first (allocations), then `passes` timed ones.  Every rank calls this (bench.py --gpus N
    `configs.config4_sharded`; the gloo tests run it on the CPU with a stand-in for the engine).

    Returns on rank 0 a dict: seconds (median pass), seconds_passes, pair_counts, per_rank (search / gather seconds and pair count of every
    rank, of the last pass), crc32_tables_match_per_rank (CRC of each rank's tables BEFORE the transfer == CRC of what rank 0 holds for that
    rank afterwards), mv / sad (the gathered tables of the last pass, pair order); on the other ranks None."""
    import zlib
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    sync = sync or (lambda: None)
    counts = pair_counts(len(pairs), world)
    mine = [pairs[p] for p in pairs_for_rank(len(pairs), rank, world)]

    def barrier():
        if world > 1 or dist.is_initialized():
            dist.barrier()

    jobs, res, mv, sad, info = [], None, None, None, None
    for i in range(passes + 1):
        res = mv = sad = None                    # the previous pass's tables go before the next pass allocates
        sync()
        barrier()
        t0 = time.perf_counter()
        res = run_share(mine)
        sync()
        t1 = time.perf_counter()
        mv, sad = gather_pair_results(res["mv"], res["sad"], len(pairs))
        sync()
        t2 = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if i:
            jobs.append(dt)
        info = {"rank": rank, "pairs": len(mine), "search_s": round(t1 - t0, 4), "gather_s": round(t2 - t1, 4), "stages": res.get("stages")}
    k = counts[rank]
    mine_crc = [zlib.crc32(res["mv"][:k].contiguous().cpu().numpy().tobytes()), zlib.crc32(res["sad"][:k].contiguous().cpu().numpy().tobytes())]
    if dist.is_initialized():
        crc, per_rank = [None] * world, [None] * world
        dist.all_gather_object(crc, mine_crc)
        dist.all_gather_object(per_rank, info)
    else:
        crc, per_rank = [mine_crc], [info]
    if rank != 0:
        return None
    crc_ok = [crc[r] == [zlib.crc32(mv[r::world].contiguous().cpu().numpy().tobytes()), zlib.crc32(sad[r::world].contiguous().cpu().numpy().tobytes())]
              for r in range(world)]
    return {"seconds": sorted(jobs)[len(jobs) // 2], "seconds_passes": jobs, "pair_counts": counts, "per_rank": per_rank,
            "crc32_tables_match_per_rank": crc_ok, "mv": mv, "sad": sad}


def search_sequence(search_pair, n_pairs, n_ctu, device):
    """run `search_pair(p, out_mv, out_sad)` (fills device tensors [n_ctu,593,2] / [n_ctu,593]) for the
    rank's pairs and gather to rank 0 (the other ranks get (None, None)).  `search_pair` is the engine call on GPUs; tests substitute a CPU stand-in."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = pairs_for_rank(n_pairs, rank, world)
    mv = torch.zeros((len(mine), n_ctu, 593, 2), dtype=torch.int16, device=device)
    sad = torch.zeros((len(mine), n_ctu, 593), dtype=torch.int32, device=device)
    for i, p in enumerate(mine):
        search_pair(p, mv[i], sad[i])
    return gather_pair_results(mv, sad, n_pairs)
