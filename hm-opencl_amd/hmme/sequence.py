"""Open-loop motion estimation of a whole sequence on one rank's share of the picture pairs (BASELINE.json config 4; driver:
tools/me_sequence.py).  Two ways to feed the GPU:

  resident   every picture the rank needs is uploaded first (a 64-picture 2160p sequence is 0.6 GB of the 288 GB), then the
             pairs are searched back to back -- the kernel-rate measurement;
  streaming  pictures arrive from a source (a YUV file, TVideoIOYuv.cpp:247/:680, or the synthetic generator) through a ring of
             plane slots: a reader thread fills page-locked host buffers, a COPY stream uploads the pictures of the next batch
             (hmme_plane_upload_async) while the COMPUTE stream searches (and refines) the current one, and a third stream
             brings the tables of finished batches back into page-locked host memory.  The library orders plane refills against
             searches that still read the old contents (include/hmme.h "Streams"); this module only has to issue a batch's
             search BEFORE the uploads that may evict its planes.

Batches: up to `pairs_per_launch` consecutive pairs of the rank's list go into ONE launch (hmme_search_pairs_device): a single
1080p pair is 510 workgroups -- less than one round of the chip's 512 workgroup slots.

The planning functions (plan_batches, plan_plane_loads) are the C++ planner of hm-opencl_amd/host/SequenceME.cpp bound with
ctypes -- one implementation for this driver and the C++ ones -- and covered by the CPU tests.
"""
import queue
import threading
import time

import numpy as np

NUM_PARTS = 593


_HOST = None


def _host_lib():
    """libhmme_host.so (hm-opencl_amd/host): the C++ host module.  The launch / plane-slot planner lives THERE
    (SequenceME.cpp plan_batches / plan_plane_loads, the C++ sequence drivers use it directly); this module binds it."""
    global _HOST
    if _HOST is None:
        import ctypes as C
        import os
        import subprocess
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(here, "host", "libhmme_host.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(here, "host")], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.hmme_host_plan.restype = C.c_int
        L.hmme_host_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_char_p, C.c_int]
        _HOST = L
    return _HOST


def _plan(pairs, pairs_per_launch, n_slots):
    import ctypes as C
    n = len(pairs)
    pr = np.ascontiguousarray(np.array(pairs, np.int32).reshape(n, 2))
    first = np.zeros(n + 1, np.int32)
    loads = np.zeros((2 * n + 2, 3), np.int32)      # a launch uploads at most the 2k pictures it reads
    where = np.zeros((n, 2), np.int32)
    nb = C.c_int(0)
    err = C.create_string_buffer(256)
    k = _host_lib().hmme_host_plan(pr.ctypes.data, n, int(pairs_per_launch), int(n_slots), first.ctypes.data, C.byref(nb), loads.ctypes.data,
                                   loads.shape[0], where.ctypes.data, err, 256)
    if k < 0:
        raise ValueError(err.value.decode())
    return first[:nb.value + 1], loads[:k], where


def plan_batches(pairs, pairs_per_launch):
    """consecutive pairs of the rank's list, at most `pairs_per_launch` (<= 16) per launch -> list of lists of indices"""
    first, _, _ = _plan(pairs, pairs_per_launch, 1 << 20)
    return [list(range(int(first[b]), int(first[b + 1]))) for b in range(len(first) - 1)]


def plan_plane_loads(pairs, batches, n_slots):
    """Which picture is uploaded into which plane slot before which batch (the C++ planner, SequenceME.cpp plan_plane_loads).

    pairs: [(cur_poc, ref_poc)] of the rank, batches: plan_batches(...).  Returns (loads, where): loads[b] = [(poc, slot)] to
    upload before batch b runs (in this order), where[b] = {poc: slot} for the pictures batch b reads.  Replacement: a slot
    whose picture is not needed by batch b is reused; among those prefer one that batch b - 1 does not read either (its refill
    then overlaps batch b - 1's search instead of waiting for it), then the one whose next use lies farthest ahead (Belady)."""
    k = max(len(b) for b in batches) if batches else 1
    first, ld, wh = _plan(pairs, k, n_slots)
    assert [list(range(int(first[b]), int(first[b + 1]))) for b in range(len(first) - 1)] == [list(b) for b in batches], "batches are not plan_batches(pairs, k)"
    loads = [[] for _ in batches]
    for b, poc, slot in ld:
        loads[int(b)].append((int(poc), int(slot)))
    where = [{} for _ in batches]
    for b, idx in enumerate(batches):
        for i in idx:
            where[b][pairs[i][0]] = int(wh[i, 0])
            where[b][pairs[i][1]] = int(wh[i, 1])
    return loads, where


class _Reader(threading.Thread):
    """fills page-locked host buffers with the pictures of the upload schedule, in order, ahead of the GPU"""

    def __init__(self, source, order, bufs):
        super().__init__(daemon=True)
        self.source, self.order, self.bufs = source, order, bufs
        self.ready = queue.Queue()             # (poc, buffer index) in schedule order
        self.free = queue.Queue()              # (buffer index, event of the upload that last read it, or None)
        for i in range(len(bufs)):
            self.free.put((i, None))
        self.read_s = 0.0
        self.error = None

    def run(self):
        try:
            for poc in self.order:
                item = self.free.get()         # handed back by the consumer only after it has ISSUED the buffer's upload
                if item is None:               # the consumer gave up (an error on its side)
                    return
                i, ev = item
                if ev is not None:
                    ev.synchronize()           # ... and that upload has run
                t0 = time.perf_counter()
                self.source.read_into(poc, self.bufs[i])
                self.read_s += time.perf_counter() - t0
                self.ready.put((poc, i))
        except Exception as e:                 # surfaced by the consumer
            self.error = e
            self.ready.put(None)


class RankResources:
    """What a pass of run_rank(stream_mode=True) allocates besides its tables -- the ring of plane slots, the page-locked host buffers, the
    streams -- kept between passes by a caller that runs several (a long-running job allocates once; hipMalloc / hipFree / hipHostMalloc
    of a pass cost ~10 ms, nothing beside a 0.3 s pass, a fifth of one rank's share of the same job on eight GPUs).  close() frees them."""

    def __init__(self):
        self.key, self.planes, self.bufs_t, self.streams = None, [], [], {}

    def close(self):
        for pl in self.planes:
            pl.close()
        self.key, self.planes, self.bufs_t = None, [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def run_rank(eng, source, pairs, width, height, bit_depth, search_range, *, stream_mode=False, pairs_per_launch=1, refine=False,
             download=False, n_slots=None, device=None, host_buffers=4, resources=None):
    """searches `pairs` [(cur_poc, ref_poc)] (this rank's share) -> dict with device tensors mv [n, n_ctu, 593, 2] int16,
    sad [n, n_ctu, 593] int32 (+ qmv / cost with refine, + host_* page-locked copies with download) and timings.
    source: .read_into(poc, out) filling a (height, width) uint8 / uint16 array (hmme.yuv.LumaFile, hmme.synth.Sequence).
    resources: a RankResources the caller keeps between passes of one geometry (streaming mode): plane slots, host buffers and streams are
    then allocated by the first pass only and stay the caller's to close."""
    import torch
    from . import api
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n = len(pairs)
    n_ctu = api.load().hmme_num_ctus(width, height)
    fp = api.FrameParams(search_range, 1, bit_depth, 0, n_ctu)
    bps = 1 if bit_depth == 8 else 2
    np_dt, t_dt = (np.uint8, torch.uint8) if bps == 1 else (np.uint16, torch.int16)
    batches = plan_batches(pairs, pairs_per_launch)
    pocs = sorted({p for pr in pairs for p in pr})
    out = {"mv": torch.zeros((n, n_ctu, NUM_PARTS, 2), dtype=torch.int16, device=dev),
           "sad": torch.zeros((n, n_ctu, NUM_PARTS), dtype=torch.int32, device=dev)}
    if refine:
        out["qmv"] = torch.zeros((n, n_ctu, NUM_PARTS, 2), dtype=torch.int16, device=dev)
        out["cost"] = torch.zeros((n, n_ctu, NUM_PARTS), dtype=torch.int32, device=dev)
    if download:
        for k in list(out):
            out["host_" + k] = torch.empty(out[k].shape, dtype=out[k].dtype, pin_memory=True)
    keep = resources if (resources is not None and stream_mode) else None
    streams = keep.streams if keep is not None else {}
    for name in ("compute",) + (("copy",) if stream_mode else ()) + (("dl",) if download else ()):
        if name not in streams:
            streams[name] = torch.cuda.Stream(device=dev)
    compute = streams["compute"]
    copy = streams["copy"] if stream_mode else compute
    dl = streams["dl"] if download else None
    stages = {"read_s": 0.0, "upload_s": 0.0, "search_s": 0.0, "refine_s": 0.0, "download_s": 0.0}
    ev_pairs = {k: [] for k in ("upload", "search", "refine", "download")}

    def timed(kind, stream):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_pairs[kind].append((a, b))
        a.record(stream)
        return b

    def launch(b, where_b):
        idx = batches[b]
        curs = [planes[where_b[pairs[i][0]]] for i in idx]
        refs = [planes[where_b[pairs[i][1]]] for i in idx]
        i0, i1 = idx[0], idx[-1] + 1
        e = timed("search", compute)
        eng.search_pairs_device(curs, refs, fp, None, out["mv"][i0:i1].data_ptr(), out["sad"][i0:i1].data_ptr(), compute.cuda_stream)
        e.record(compute)
        if refine:
            e = timed("refine", compute)
            eng.refine_pairs_device(curs, refs, fp, None, out["mv"][i0:i1].data_ptr(), 1, out["qmv"][i0:i1].data_ptr(),
                                    out["cost"][i0:i1].data_ptr(), compute.cuda_stream)
            e.record(compute)
        if download:
            done = torch.cuda.Event()
            done.record(compute)
            dl.wait_event(done)
            e = timed("download", dl)
            with torch.cuda.stream(dl):
                for k in ("mv", "sad") + (("qmv", "cost") if refine else ()):
                    out["host_" + k][i0:i1].copy_(out[k][i0:i1], non_blocking=True)
            e.record(dl)

    planes = []
    reader = None
    try:
        if not stream_mode:
            # ---- resident: one plane per picture, uploaded before the clock starts
            where_all = {}
            host = np.empty((height, width), np_dt)
            for p in pocs:
                pl = eng.plane(width, height, bit_depth)
                source.read_into(p, host)
                if bps == 1:
                    pl.upload_u8(host)
                else:
                    eng._check(eng.L.hmme_plane_upload_pel(pl.h, host.ctypes.data, width))
                where_all[p] = len(planes)
                planes.append(pl)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for b in range(len(batches)):
                launch(b, where_all)
        else:
            # ---- streaming: ring of plane slots, uploads one batch ahead of the search
            k = max(len(b) for b in batches) if batches else 1
            n_slots = n_slots or max(8, 2 * k + 2)
            loads, where = plan_plane_loads(pairs, batches, n_slots)
            n_planes = min(n_slots, max(1, len(pocs)))
            key = (id(eng), width, height, bit_depth, n_planes, host_buffers)
            if keep is not None and keep.key == key:
                planes, bufs_t = keep.planes, keep.bufs_t
            else:
                if keep is not None:
                    keep.close()
                planes = [eng.plane(width, height, bit_depth) for _ in range(n_planes)]
                bufs_t = [torch.empty((height, width), dtype=t_dt, pin_memory=True) for _ in range(host_buffers)]
                if keep is not None:
                    keep.key, keep.planes, keep.bufs_t = key, planes, bufs_t
            bufs = [t.numpy().view(np_dt) for t in bufs_t]
            order = [p for l in loads for (p, _) in l]
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            reader = _Reader(source, order, bufs)
            reader.start()

            def upload(b):
                for poc, slot in loads[b]:
                    item = reader.ready.get()
                    if item is None:
                        raise reader.error
                    got, i = item
                    assert got == poc
                    e = timed("upload", copy)
                    planes[slot].upload_async(bufs[i].ctypes.data, width, bps, copy.cuda_stream)
                    e.record(copy)
                    reader.free.put((i, e))     # the reader waits for this upload before it overwrites host buffer i

            if batches:
                upload(0)
            for b in range(len(batches)):
                launch(b, where[b])             # issued first: the uploads below wait for it only where they evict its planes
                if b + 1 < len(batches):
                    upload(b + 1)
            reader.join()
            stages["read_s"] = reader.read_s
            eng.upload_status(copy.cuda_stream)
        torch.cuda.synchronize(dev)
        out["seconds"] = time.perf_counter() - t0
        for kind, lst in ev_pairs.items():
            stages[kind + "_s"] = sum(a.elapsed_time(b) for a, b in lst) * 1e-3
        out["stages"] = {k: round(v, 5) for k, v in stages.items()}
        out["launches"] = len(batches)
        out["plane_slots"] = len(planes)
        out["uploads"] = len(ev_pairs["upload"]) if stream_mode else len(pocs)
    finally:
        if reader is not None and reader.is_alive():
            reader.free.put(None)              # unblocks a reader that waits for a buffer after an error on this side
        torch.cuda.synchronize(dev)
        if keep is None or keep.planes is not planes:
            for pl in planes:
                pl.close()
    return out
