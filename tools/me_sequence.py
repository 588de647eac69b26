#!/usr/bin/env python3
"""Open-loop motion estimation of a whole sequence, picture pairs sharded across GPUs
(BASELINE.json config 4: 3840x2160, 64 pictures, randomaccess GOP, 8 x MI355X).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        tools/me_sequence.py --frames 64 --gop randomaccess --size 2160p [--yuv file.yuv]

Every rank keeps the pictures it needs resident in HBM (a 64-picture 2160p sequence is 0.65 GB of the
288 GB), searches pairs p = rank, rank+N, ... and the 593-entry tables of all pairs are gathered with one
RCCL all-gather per table (hmme/shard.py).  Rank 0 prints a JSON summary.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--gop", default="randomaccess", choices=["randomaccess", "lowdelay_P"])
    ap.add_argument("--size", default="2160p")
    ap.add_argument("--search-range", type=int, default=64)
    ap.add_argument("--yuv", default=None, help="planar 8-bit 4:2:0 file; synthetic frames if omitted")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from hmme import api, shard, synth, yuv

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    w, h = {"2160p": (3840, 2160), "1080p": (1920, 1080)}.get(args.size) or tuple(int(v) for v in args.size.split("x"))
    pairs = shard.gop_pairs(args.frames, args.gop)
    mine = [pairs[p] for p in shard.pairs_for_rank(len(pairs), rank, world)]
    eng = api.Engine(local, 64)
    eng.set_lambda(57.9)
    planes = {}
    for poc in sorted({f for pr in mine for f in pr}):
        pl = eng.plane(w, h)
        if args.yuv:
            pl.upload_u8(yuv.read_luma(args.yuv, w, h, poc))
        else:   # frame poc = texture translated by a per-frame global motion (synthetic sequence)
            cur, _, _ = synth.make_pair(w, h, seed=777, max_mv=0, noise_sigma=0.0, shift=(3 * poc, 2 * poc),
                                        pad=3 * args.frames + 4)
            pl.upload_pel(cur, (synth.MARGIN, synth.MARGIN))
        planes[poc] = pl
    n_ctu = api.load().hmme_num_ctus(w, h)
    fp = api.FrameParams(args.search_range, 1, 8, 0, n_ctu)
    stream = torch.cuda.current_stream().cuda_stream

    def search_pair(p, out_mv, out_sad):
        cur_poc, ref_poc = pairs[p]
        eng.search_frame_device(planes[cur_poc], planes[ref_poc], fp, None, out_mv.data_ptr(), out_sad.data_ptr(), stream)

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    mv, sad = shard.search_sequence(search_pair, len(pairs), n_ctu, dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        m = mv[:, :, 592].to(torch.int32)
        print(json.dumps({"pairs": len(pairs), "gpus": world, "seconds": round(dt, 4),
                          "pairs_per_s": round(len(pairs) / dt, 2), "ctus_per_s": round(len(pairs) * n_ctu / dt, 1),
                          "median_mv_64x64_of_first_pairs": [[int(m[i, :, 0].median()), int(m[i, :, 1].median())] for i in range(min(4, len(pairs)))],
                          "first_pairs": pairs[:4]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
