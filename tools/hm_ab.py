#!/usr/bin/env python3
"""A/B run of the reference's own encoder (oracle/_ref/TAppEncoder_hmme, built by `make -C oracle dropin`)
with its CPU searches and with the HIP engine behind TEncOpenCL: bits / PSNR / wall time per configuration.
SURVEY.md 8f row 4.  Usage: python tools/hm_ab.py [--size 416x240] [--frames 5] [--search-range 64]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
from hmme import synth, yuv  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_hmme")
EXE_HM = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_hmme_hm")   # with tools/hm_patch applied
CFGS = {"P": os.path.join(ROOT, "tests", "hm", "lowdelay_P_small.cfg"), "B": os.path.join(ROOT, "tests", "hm", "lowdelay_B_small.cfg"),
        "P4": os.path.join(ROOT, "tests", "hm", "lowdelay_P4_small.cfg"), "RA": os.path.join(ROOT, "tests", "hm", "randomaccess_small.cfg")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="416x240")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--search-range", type=int, default=64)
    ap.add_argument("--gop", default="P", choices=sorted(CFGS), help="low-delay P (one reference), B (two references, bi-prediction), P4 (four references) or RA (hierarchical B, GOP 8)")
    ap.add_argument("--skip-full-search", action="store_true", help="leave out the CPU exhaustive search (minutes per picture at 1080p)")
    ap.add_argument("--log", default=None, help="append progress lines to this file as configurations start and finish (long runs on the "
                                                "GPU box must keep writing under gpurun_out/)")
    ap.add_argument("--hm-args", default="", help="extra TAppEncoder arguments for every configuration, e.g. '--Profile=main10 --InternalBitDepth=10'")
    ap.add_argument("--only", default=None, help="run only the configurations whose name contains this text (e.g. GPU_FRAC)")
    ap.add_argument("--fade", type=float, default=0.0, help="fade the clip to black by this fraction per picture (what explicit weighted prediction is for; "
                                                            "combine with --hm-args '--WeightedPredP=1 --WeightedPredB=1')")
    ap.add_argument("--verify", action="store_true", help="HMME_VERIFY=1 on the patched encoder (slower: runs HM's xPatternSearch beside the engine)")
    args = ap.parse_args()
    CFG = CFGS[args.gop]
    w, h = (int(v) for v in args.size.split("x"))
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "in.yuv")
    pics = []
    for t in range(args.frames):   # textured content with per-region motion that grows over time + noise
        cur, _, _ = synth.make_pair(w, h, seed=11, max_mv=2, region=96, noise_sigma=1.5, shift=(3 * t, -2 * t),
                                    pad=3 * args.frames + 8, margin=0)
        if args.fade:
            cur = np.clip(np.rint(cur * max(0.0, 1.0 - args.fade * t)), 0, 255)
        pics.append(cur.astype(np.uint8))
    yuv.write_luma_420(src, pics)
    rows = []
    for name, exe, extra in (("CPU TZ search (FastSearch=1)", EXE, ["--OpenCL=0", "--FastSearch=1"]),
                             ("CPU full search (FastSearch=0)", EXE, ["--OpenCL=0", "--FastSearch=0"]),
                             ("hmme, reference call sites (ME_MODE_OCL_COMPAT)", EXE, ["--OpenCL=1", "--FastSearch=1", "--KernelOpenCL=embedded"]),
                             ("hmme, tools/hm_patch (ME_MODE_HM, bi-pred tables, edge CTUs)", EXE_HM, ["--OpenCL=1", "--FastSearch=1", "--KernelOpenCL=embedded"]),
                             ("hmme, tools/hm_patch + HMME_GPU_FRAC=1 (refinement tables too)", EXE_HM, ["--OpenCL=1", "--FastSearch=1", "--KernelOpenCL=embedded"])):
        if args.skip_full_search and "--FastSearch=0" in extra:
            continue
        if args.only and args.only not in name:
            continue
        if args.log:
            with open(args.log, "a") as f:
                f.write(f"{time.strftime('%H:%M:%S')} start: {name}\n")
        t0 = time.time()
        env = dict(os.environ, HMME_TRACE="1")
        if args.verify and exe == EXE_HM:
            env["HMME_VERIFY"] = "1"
        if "GPU_FRAC" in name:
            env["HMME_GPU_FRAC"] = "1"
        r = subprocess.run([exe, "-c", CFG, "-i", src, "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(args.frames),
                            f"--SearchRange={args.search_range}", "-b", os.path.join(tmp, "s.bin"), *extra, *args.hm_args.split()],
                           capture_output=True, text=True, env=env, cwd=tmp)
        dt = time.time() - t0
        if args.log:
            with open(args.log, "a") as f:
                f.write(f"{time.strftime('%H:%M:%S')} done in {dt:.1f} s: {name}\n")
        if r.returncode != 0:
            rows.append({"config": name, "error": r.stderr[-300:]})
            continue
        pocs = re.findall(r"POC\s+(\d+).*?(\d+) bits \[Y ([0-9.]+) dB", r.stdout)
        p_bits = sum(int(b) for p, b, y in pocs if int(p) > 0)
        p_psnr = float(np.mean([float(y) for p, b, y in pocs if int(p) > 0]))
        m = re.search(r"(\d+) calcMotionVectors calls, (\d+) failed, (\d+) edge-CTU, (\d+) bi-pred, (\d+) results verified against xPatternSearch, (\d+) differ", r.stderr)
        me = re.search(r"differ, ([0-9.]+) s inside the engine calls \((\d+) weighted\)", r.stderr)
        # HM's own clock per picture ("[ET   12 ]", whole seconds) and for the run ("Total Time:   123.456 sec.")
        et = [int(v) for v in re.findall(r"\[ET\s+(\d+)\s*\]", r.stdout)]
        tt = re.search(r"Total Time:\s+([0-9.]+) sec", r.stdout)
        row = {"config": name, "inter_bits": p_bits, "inter_psnr_y": round(p_psnr, 3), "wall_s": round(dt, 2),
               "hm_total_time_s": float(tt.group(1)) if tt else None, "hm_et_per_picture_s": et,
               "timing_run": not (args.verify and exe == EXE_HM)}
        if m and int(m.group(1)):
            row.update(dict(zip(("engine_calls", "failed", "edge_ctu_calls", "bipred_calls", "verified", "verify_mismatches"), (int(v) for v in m.groups()))))
            if me:   # wall time of the engine calls as the encoder saw them (TEncOpenCL's own clock), and what is left for the rest of HM
                row["engine_seconds"] = float(me.group(1))
                row["engine_ms_per_call"] = round(1e3 * float(me.group(1)) / int(m.group(1)), 4)
                row["engine_share_of_wall"] = round(float(me.group(1)) / dt, 5)
                row["weighted_calls"] = int(me.group(2))
        rows.append(row)
    print(json.dumps({"clip": f"{w}x{h} x {args.frames} synthetic", "verify": bool(args.verify),
                      "note": "wall_s of a run with verify = true includes HM's CPU full search beside every engine call: a correctness run, not a timing" if args.verify else
                              "timing run: HMME_VERIFY off", "gop": {"P": "low-delay P", "B": "low-delay B", "P4": "low-delay P, 4 references", "RA": "random access, GOP 8"}[args.gop], "search_range": args.search_range, "hm_args": args.hm_args, "fade_per_picture": args.fade, "runs": rows}, indent=1))


if __name__ == "__main__":
    main()
