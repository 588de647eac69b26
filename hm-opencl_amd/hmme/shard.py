"""Frame sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the MI355X node, "gloo" in the CPU tests).

Open-loop motion estimation over a sequence is embarrassingly parallel per (current, reference)
picture pair: pair p is searched by rank p % world with no exchange during the
search.  The only collective is the result gather: every rank contributes an equally sized
[pairs_per_rank, n_ctu, 593] block of (mv, sad) -- 9.7 MB per 2160p pair -- through
all_gather_into_tensor, and the blocks are re-interleaved into pair order.
"""
import torch
import torch.distributed as dist


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
    """every rank contributes the same number of blocks to the gather (short ranks pad)"""
    return (n_pairs + world - 1) // world


def gather_pair_results(local_mv, local_sad, n_pairs):
    """local_mv: [k, n_ctu, 593, 2] int16, local_sad: [k, n_ctu, 593] int32 with k = pairs_per_rank
    (rows beyond the rank's real pairs are padding).  Returns (mv, sad) in pair order on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local_mv[:n_pairs], local_sad[:n_pairs]
    k = pairs_per_rank(n_pairs, world)
    assert local_mv.shape[0] == k and local_sad.shape[0] == k
    # an (mvx, mvy) int16 pair travels as one int32 word (TComMv is 4 bytes; gloo has no int16 collectives)
    mv_shape = tuple(local_mv.shape)
    local_mv = local_mv.contiguous().view(torch.int32)
    # concatenated layout (world * k, ...): accepted by both the RCCL and the gloo implementation
    dev = local_mv.device
    if dist.get_backend() == "gloo" and dev.type != "cpu":   # rehearsal of the N > 1 path without RCCL: stage on host
        local_mv, local_sad = local_mv.cpu(), local_sad.cpu()
    g_mv = torch.empty((world * k,) + tuple(local_mv.shape[1:]), dtype=local_mv.dtype, device=local_mv.device)
    g_sad = torch.empty((world * k,) + tuple(local_sad.shape[1:]), dtype=local_sad.dtype, device=local_sad.device)
    dist.all_gather_into_tensor(g_mv, local_mv.contiguous())
    dist.all_gather_into_tensor(g_sad, local_sad.contiguous())
    g_mv, g_sad = g_mv.to(dev), g_sad.to(dev)
    g_mv = g_mv.view((world, k) + tuple(local_mv.shape[1:]))
    g_sad = g_sad.view((world, k) + tuple(local_sad.shape[1:]))
    # block [r, i] holds pair i * world + r
    mv = g_mv.transpose(0, 1).reshape((k * world,) + tuple(local_mv.shape[1:])).view(torch.int16)
    mv = mv.reshape((k * world,) + mv_shape[1:])[:n_pairs]
    sad = g_sad.transpose(0, 1).reshape((k * world,) + tuple(local_sad.shape[1:]))[:n_pairs]
    return mv, sad


def gather_packed(buf, out=None, async_op=False):
    """one collective for both result tables.  buf: int32 [2, k, n_ctu, 593] (plane 0 = TComMv words,
    plane 1 = SADs) of this rank; returns (out int32 [world, 2, k, n_ctu, 593], work-or-None).
    With async_op=True the gather runs on RCCL's stream while the caller launches the next search
    (the handle's wait() is a stream-side dependency, not a host block)."""
    if not dist.is_initialized():
        return buf.unsqueeze(0), None
    world = dist.get_world_size()   # a world of 1 still goes through the collective (bench.py's one-GPU rehearsal of the RCCL calls)
    if out is None:
        out = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
    flat_out = out.view((world * buf.shape[0],) + tuple(buf.shape[1:]))
    if dist.get_backend() == "gloo" and buf.device.type != "cpu":   # rehearsal path: stage on the host
        tmp = torch.empty(flat_out.shape, dtype=buf.dtype)
        dist.all_gather_into_tensor(tmp, buf.cpu())
        flat_out.copy_(tmp)
        return out, None
    work = dist.all_gather_into_tensor(flat_out, buf, async_op=async_op)
    return out, work


class PipelinedGather:
    """Double-buffered result exchange of a stream of search steps (bench.py --gpus N, an open-loop ME pass).

    Step k writes its tables into local buffer k % 2 and starts their all-gather; the gather runs on the collective
    backend's own stream / thread while step k+1 searches into the other buffer, so the 9.7 MB per rank and 2160p
    pair never stall the search kernel.  Before buffer b is written again (step k+2) the gather that read it (step k)
    is waited for -- with RCCL a stream-side dependency, not a host block -- and only then is its gathered block
    handed to `consume` and recycled.

    make_local() -> tensor            this rank's [2, k, n_ctu, 593] int32 block (plane 0 TComMv words, plane 1 SADs)
    launch(buf, step)                 enqueue the search of step `step` writing into buf (engine call; tests: a stand-in)
    consume(step, gathered) or None   gathered = [world, ...] block of `step`, valid only during the call
    """

    def __init__(self, make_local, distributed, async_op=True):
        self.world = dist.get_world_size() if (distributed and dist.is_initialized()) else 1
        self.distributed = bool(distributed)
        self.async_op = async_op
        self.bufs = [make_local(), make_local()]
        self.gathered = [torch.empty((self.world,) + tuple(b.shape), dtype=b.dtype, device=b.device) if self.distributed else None
                         for b in self.bufs]
        self.pending = [None, None]     # (step, work-or-None) of the gather that last used buffer b
        self.k = 0

    def _retire(self, b, consume):
        if self.pending[b] is None:
            return
        step, work = self.pending[b]
        if work is not None:
            work.wait()
        self.pending[b] = None
        if consume is not None:
            consume(step, self.gathered[b] if self.distributed else self.bufs[b].unsqueeze(0))

    def step(self, launch, consume=None):
        b = self.k & 1
        self._retire(b, consume)        # the gather that read bufs[b] (two steps ago) is done before bufs[b] is overwritten
        launch(self.bufs[b], self.k)
        work = None
        if self.distributed:   # the one exchange step of the path: tables of all `world` pairs to every rank
            _, work = gather_packed(self.bufs[b], self.gathered[b], async_op=self.async_op)
        self.pending[b] = (self.k, work)
        self.k += 1
        return b

    def drain(self, consume=None):
        for b in ((self.k & 1), (self.k & 1) ^ 1):   # oldest first
            self._retire(b, consume)

    @property
    def last_local(self):
        """this rank's buffer of the most recent step"""
        return self.bufs[(self.k - 1) & 1]


def search_sequence(search_pair, n_pairs, n_ctu, device):
    """run `search_pair(p, out_mv, out_sad)` (fills device tensors [n_ctu,593,2] / [n_ctu,593]) for the
    rank's pairs and gather.  `search_pair` is the engine call on GPUs; tests substitute a CPU stand-in."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    k = pairs_per_rank(n_pairs, world)
    mv = torch.zeros((k, n_ctu, 593, 2), dtype=torch.int16, device=device)
    sad = torch.zeros((k, n_ctu, 593), dtype=torch.int32, device=device)
    for i, p in enumerate(pairs_for_rank(n_pairs, rank, world)):
        search_pair(p, mv[i], sad[i])
    return gather_pair_results(mv, sad, n_pairs)
