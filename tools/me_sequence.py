#!/usr/bin/env python3
"""Open-loop motion estimation of a whole sequence, picture pairs sharded across GPUs
(BASELINE.json config 4: 3840x2160, 64 pictures, randomaccess GOP, 8 x MI355X).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        tools/me_sequence.py --frames 64 --gop randomaccess --size 2160p [--yuv file.yuv] [--bit-depth 10]
                             [--stream] [--pairs-per-launch K] [--refine] [--download] [--dump out.npz]

Rank r searches pairs r, r+N, ... (hmme/shard.py); the 593-entry tables of all pairs are gathered to rank 0 over RCCL (grouped
send / receive, one batch per table).  Default: every picture the rank needs is resident in HBM before the clock starts (a 64-picture 2160p sequence is 0.6 GB
of the 288 GB).  --stream: pictures come from the file (or the generator) through a ring of plane slots while the GPU searches
(hmme/sequence.py): reader thread -> page-locked buffers -> copy stream || compute stream || download stream; the clock then
includes reading and uploading.  --pairs-per-launch K puts K picture pairs into one launch (small pictures do not fill the
chip one pair at a time).  --dump writes tables (a subset with --dump-pairs / --dump-ctus) for the parity tests; this tool
itself never touches the CPU oracle.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--gop", default="randomaccess", choices=["randomaccess", "lowdelay_P"])
    ap.add_argument("--size", default="2160p")
    ap.add_argument("--search-range", type=int, default=64)
    ap.add_argument("--bit-depth", type=int, default=8, help="8: 8-bit file / planes; 9..12: 16-bit little-endian file, u16 planes")
    ap.add_argument("--yuv", default=None, help="planar 4:2:0 file (8-bit samples, or 16-bit LE words with --bit-depth > 8); synthetic pictures if omitted")
    ap.add_argument("--chroma", default="420", choices=["400", "420", "422", "444"])
    ap.add_argument("--stream", action="store_true", help="stream pictures through a ring of plane slots instead of uploading all first")
    ap.add_argument("--slots", type=int, default=0, help="plane slots of the ring (--stream); default max(8, 2K + 2)")
    ap.add_argument("--pairs-per-launch", type=int, default=1)
    ap.add_argument("--refine", action="store_true", help="also run the fractional-pel refinement of every pair (hmme_refine_pairs_device)")
    ap.add_argument("--download", action="store_true", help="bring each batch's tables into page-locked host memory on a third stream")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered tables (npz) for the parity tests")
    ap.add_argument("--dump-pairs", default=None, help="comma-separated pair indices to dump (default all)")
    ap.add_argument("--dump-ctus", default=None, help="first:count CTU range to dump (default all)")
    ap.add_argument("--repeat", type=int, default=1, help="run the pass this many times, report the last (the first includes allocations)")
    ap.add_argument("--rank-timeout", type=float, default=600.0,
                    help="N > 1: seconds a rank may take from its start to the end of its first barrier (rendezvous, RCCL communicator) before it "
                         "gives up with exit code 3, naming itself and the stage it hung in; 0 = no limit")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from hmme import api, sequence, shard, synth, yuv

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or (os.environ.get("HMME_SEQ_FORCE_DIST") == "1" and "RANK" in os.environ)   # world 1: rehearses the RCCL gather
    if use_dist:
        # the same start-up evidence and the same loud failure as bench.py --gpus N (hmme/shard.py): what this rank sees before rendezvous, the
        # ranks counted through the job's store before the first collective, a watchdog that ends a rank stuck before its first barrier
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        watchdog = shard.StartupWatchdog(rank, world, args.rank_timeout, label="me_sequence.py")
        props = torch.cuda.get_device_properties(local)
        sys.stderr.write(f"me_sequence.py: rank {rank}/{world} pid {os.getpid()} local_rank {local}: hipGetDeviceCount {torch.cuda.device_count()}, device {props.name} "
                         f"pci {getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}.0 uuid {getattr(props, 'uuid', '')}, "
                         f"RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())}, MASTER {os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')}\n")
        sys.stderr.flush()
        watchdog.stage("init_process_group (rendezvous)")
        dist.init_process_group("nccl", device_id=dev)
        watchdog.stage("rendezvous store count")
        seen = shard.rendezvous_report(rank, world, timeout_s=max(10.0, args.rank_timeout / 2) if args.rank_timeout > 0 else 120.0)
        if rank == 0 and seen is not None and seen != world:
            raise SystemExit(f"me_sequence.py: only {seen} of {world} ranks reached the rendezvous store: nothing reported")
        watchdog.stage("first barrier (RCCL communicator)")
        dist.barrier()
        torch.cuda.synchronize()
        watchdog.done()
    w, h = {"2160p": (3840, 2160), "1080p": (1920, 1080), "720p": (1280, 720)}.get(args.size) or tuple(int(v) for v in args.size.split("x"))
    bd = args.bit_depth
    pairs = shard.gop_pairs(args.frames, args.gop)
    mine = [pairs[p] for p in shard.pairs_for_rank(len(pairs), rank, world)]
    eng = api.Engine(local, max(64, args.search_range))
    eng.set_lambda(57.9)
    if args.yuv:
        source = yuv.LumaFile(args.yuv, w, h, 8 if bd == 8 else 16, args.chroma)
    else:   # picture t = one texture translated by (3t, 2t)
        source = synth.Sequence(w, h, args.frames, seed=777, bit_depth=bd)
    n_ctu = api.load().hmme_num_ctus(w, h)
    res = None
    for _ in range(max(1, args.repeat)):
        res = None   # frees the previous pass's tables before the next allocates
        if use_dist:
            dist.barrier()
        res = sequence.run_rank(eng, source, mine, w, h, bd, args.search_range, stream_mode=args.stream,
                                pairs_per_launch=args.pairs_per_launch, refine=args.refine, download=args.download,
                                n_slots=args.slots or None, device=dev)
    dt = res["seconds"]
    if use_dist:   # the slowest rank's time is the job's
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the one exchange step: every rank's tables to rank 0, which puts them into pair order (hmme/shard.py gather_to_root: grouped
    # point-to-point transfers; a rank with one pair fewer simply sends one table fewer)
    import time
    t0 = time.perf_counter()
    mv, sad = shard.gather_pair_results(res["mv"], res["sad"], len(pairs))
    if args.refine:
        qmv, cost = shard.gather_pair_results(res["qmv"], res["cost"], len(pairs))
    torch.cuda.synchronize()
    gather_s = time.perf_counter() - t0
    if rank == 0:
        sads = 0   # 4x4-block SAD evaluations of one picture search (clipped windows at the picture edges)
        for cy in range(0, h, 64):
            for cx in range(0, w, 64):
                ltx, lty, rbx, rby = api.set_search_range(0, 0, args.search_range, cx, cy, w, h)
                sads += (min(64, w - cx) // 4) * (min(64, h - cy) // 4) * (rbx - ltx + 1) * (rby - lty + 1)
        m = mv[:, :, 592].to(torch.int32)
        med = torch.stack([m[:, :, 0].median(dim=1).values, m[:, :, 1].median(dim=1).values], dim=1).cpu().tolist()
        total = dt + gather_s
        print(json.dumps({"pairs": len(pairs), "gpus": world, "seconds": round(dt, 4), "gather_seconds": round(gather_s, 4),
                          "pairs_per_s": round(len(pairs) / total, 2), "ctus_per_s": round(len(pairs) * n_ctu / total, 1),
                          "gsad_per_s": round(len(pairs) * sads / total / 1e9, 1),
                          "mode": {"stream": bool(args.stream), "pairs_per_launch": args.pairs_per_launch, "refine": bool(args.refine),
                                   "download": bool(args.download), "bit_depth": bd, "source": args.yuv or "synthetic",
                                   "collective": "gather to rank 0: grouped send/recv (nccl), world %d" % world if use_dist else "none (one rank)",
                                   "pairs_per_rank": shard.pair_counts(len(pairs), world)},
                          "size": [w, h], "search_range": args.search_range,
                          "rank0": {"pairs": len(mine), "launches": res["launches"], "plane_slots": res["plane_slots"], "uploads": res["uploads"],
                                    "stages": res["stages"]},
                          "median_mv_64x64": med, "pair_list": pairs,
                          "median_mv_64x64_of_first_pairs": med[:4], "first_pairs": pairs[:4]}))
        if args.dump:
            sel = [int(v) for v in args.dump_pairs.split(",")] if args.dump_pairs else list(range(len(pairs)))
            c0, cn = (int(v) for v in args.dump_ctus.split(":")) if args.dump_ctus else (0, n_ctu)
            d = {"pairs": np.array([pairs[i] for i in sel], np.int32), "pair_index": np.array(sel, np.int32), "ctu_first": c0,
                 "mv": mv[sel, c0:c0 + cn].cpu().numpy(), "sad": sad[sel, c0:c0 + cn].cpu().numpy().view(np.uint32)}
            if args.refine:
                d["qmv"] = qmv[sel, c0:c0 + cn].cpu().numpy()
                d["cost"] = cost[sel, c0:c0 + cn].cpu().numpy().view(np.uint32)
            if args.download and world == 1:   # what the download stream delivered must equal the device tables
                d["host_equal"] = bool(torch.equal(res["host_mv"], res["mv"].cpu()) and torch.equal(res["host_sad"], res["sad"].cpu()))
            np.savez(args.dump, **d)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
