#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hmme engine (BASELINE.json): integer full-search motion
estimation of a 3840x2160 8-bit frame against one reference picture, SearchRange 64, CTU 64,
FEN 1 (encoder_lowdelay_P_main.cfg), all 593 PU shapes per CTU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 2160p|1080p]

A *step* = one whole-picture search (2040 CTUs at 2160p) on every rank; ranks hold different
frames of the sequence (frame sharding, no data-path collective) and the per-step results are
gathered to rank 0 with RCCL (torch.distributed "nccl": grouped ncclSend / ncclRecv) when N > 1.
Inputs are resident in HBM before the timed region; the timed region is K steps between
barrier + synchronize, max over ranks.  Rank 0 prints ONE JSON line: metric GSAD/s (4x4-block SAD
evaluations per second, whole job), plus `roofline` (algorithmic bytes / measured kernel time
against 8 TB/s) and `cpu_baseline` (the CPU oracle's exhaustive search and HM's xTZSearch
restatement timed on this node's cores).

N > 1: under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is one rank.
Invoked plainly as `python bench.py --gpus N` it starts the N ranks itself -- a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...` launched BEFORE anything in this
process touches the GPU -- and relays rank 0's line and the exit code.  The line then carries the
evidence that N devices took part: `ranks_seen`, every rank's device (PCI bus id; N distinct),
per-rank kernel and step times, the gather's time and bytes, and a cross-rank check (CRC of every
rank's tables before the gather == CRC of the block rank 0 received; CTUs of the last rank against
the CPU oracle), `multi_gpu.compute_only` (the same K steps with the exchange switched off) and
`multi_gpu.exchange_cost` (the difference, rank by rank), and `configs.config4_sharded`: BASELINE
config 4's 124-pair random-access GOP dealt pair p -> rank p mod N, streamed, gathered, with the pair
counts, the same job on rank 0 alone (strong scaling) and a pair of the last rank against the oracle.

At N = 1 the line also times and verifies, after the headline and never inside `value`, the other
single-GPU BASELINE configurations (`configs`: 1080p SR 64; 2160p 10-bit SR 128; the 64-picture
random-access sequence streamed through one GPU; 720p and 1440p, pictures that do not fill whole rounds
of the chip's workgroup slots) and the refinement kernel on coherent, mixed and unrelated content
(`refine`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))

SIZES = {"2160p": (3840, 2160), "1080p": (1920, 1080), "1440p": (2560, 1440), "1600p": (2560, 1600), "720p": (1280, 720), "1200p": (1920, 1200)}


def algo_bytes_per_ctu(sr, bit_depth):
    """B_px * (64*64 + (64+2SR)^2) + 593*8  (CTU + window + results)"""
    return (1 if bit_depth == 8 else 2) * (64 * 64 + (64 + 2 * sr) ** 2) + 593 * 8


HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LAMBDA = 57.9                                          # fixed, recorded


def work_4x4_sads(api, w, h, sr, pred=None):
    """4x4-block SAD evaluations of one picture search: sum over CTUs of (in-picture 4x4 blocks) x
    (candidates of the CTU's window: xSetSearchRange around the CTU's predictor, clipped by clipMv -- the candidates actually searched)"""
    total = 0
    ctu = 0
    for cy in range(0, h, 64):
        for cx in range(0, w, 64):
            px, py = (int(pred[ctu, 0]), int(pred[ctu, 1])) if pred is not None else (0, 0)
            ltx, lty, rbx, rby = api.set_search_range(px, py, sr, cx, cy, w, h)
            blocks = (min(64, w - cx) // 4) * (min(64, h - cy) // 4)
            total += blocks * (rbx - ltx + 1) * (rby - lty + 1)
            ctu += 1
    return total


def warm_clock(eng, planes, fp, buf, stream, launches=24):
    """untimed launches in front of a timed leg: the power management takes ~10 launches (tens of ms of load) to bring a GPU that idled
    through the host work before the leg (picture synthesis, oracle checks) back to its sustained clock; a leg of a few short launches
    would otherwise be timed on the way up (the same set-up as the headline's, main())"""
    for _ in range(launches):
        eng.search_pairs_device([planes[0]], [planes[1]], fp, None, buf[0].data_ptr(), buf[1].data_ptr(), stream)


def library_build_id():
    """hmme_build_id() of the library that is actually LOADED (HMME_LIB variants included): a hash over the kernel sources and the
    flags that shape them, compiled into libhmme.so -- ties a committed counter summary to the binary that was profiled"""
    from hmme import api
    return api.build_id()


def profile_label(size, sr, bd):
    return f"{size}_sr{sr}" + ("" if bd == 8 else f"_{bd}bit")


def pmc_profile(size, sr, bd):
    """derived figures of the last rocprofv3 --pmc passes of this same command (tools/profile_bench.sh), committed under
    profiles/ -- counters cannot be read from inside the timed process.  {} when there is no summary for this
    configuration or when it was taken from other kernel sources than the ones this library was built from."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", f"latest_pmc_{profile_label(size, sr, bd)}.json")))
    except (OSError, ValueError):
        return {}
    if d.get("library_build_id") != library_build_id() or library_build_id() in ("", "unknown"):   # a library without an id proves nothing
        return {"stale": True}
    return d


LIVE_PASSES = ("FETCH_SIZE", "WRITE_SIZE",
               "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
               "GRBM_GUI_ACTIVE GRBM_COUNT")


def live_pmc(extra_args, kernel_names, budget_s=150.0):
    """The counters of THIS run: after the timed region the benchmark starts `rocprofv3 --kernel-trace --pmc <one pass>` over a short
    headline-only run of itself (child processes; each counter group in a pass of its own, never combined with other trace domains, as
    MI355X_MICROARCH.md prescribes) and condenses the per-dispatch means of the kernels named (the search kernel; the refinement kernel
    of the `refine` leg of the same child run).  {} when rocprofv3 is not there or a pass fails -- the line then falls back to the
    committed summary of the same library (pmc_profile).  -> {kernel name: figures}"""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return {}
    tmp = tempfile.mkdtemp(prefix="hmme_pmc_")
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    means = {k: {} for k in kernel_names}
    kns = {k: {} for k in kernel_names}
    scratch = {}
    t0 = time.time()
    try:
        for i, group in enumerate(LIVE_PASSES):
            if time.time() - t0 > budget_s:
                return {}
            out = os.path.join(tmp, f"pass{i}")
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *group.split(), "--output-format", "csv", "-d", out, "--",
                                sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra_args],
                               capture_output=True, text=True, timeout=120, env=env, cwd=tmp)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {}
            acc = {k: collections.defaultdict(list) for k in kernel_names}
            dur = {k: {} for k in kernel_names}
            for row in csv.DictReader(open(files[0])):
                for k in kernel_names:
                    if k + "<" in row["Kernel_Name"]:
                        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        dur[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                        for col in ("Scratch_Size", "Private_Segment_Size", "Scratch_Memory_Size"):
                            if row.get(col) not in (None, ""):
                                scratch[k] = max(scratch.get(k, 0), int(float(row[col])))
            if not acc[kernel_names[0]]:
                return {}
            for k in kernel_names:
                for c, v in acc[k].items():
                    means[k][c] = sum(v) / len(v)
                    kns[k][c] = sum(dur[k].values()) / len(dur[k])
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k in kernel_names:
        m, n = means[k], kns[k]
        if not m:
            continue
        d = {"seconds": round(time.time() - t0, 1), "passes": list(LIVE_PASSES)}
        if k in scratch:
            d["scratch_bytes_per_lane"] = scratch[k]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:   # KiB; on gfx950 FETCH_SIZE reports half of a coalesced stream (the guide's correction)
            d["hbm_traffic_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024
            d["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
            clk = m["GRBM_GUI_ACTIVE"] / 8 / n["GRBM_GUI_ACTIVE"]          # GHz: the counter sums over the 8 XCDs
            d.update({"valu_wave_instructions_per_launch": m["SQ_INSTS_VALU"], "effective_clock_ghz": clk,
                      "valu_busy_frac": m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * n["SQ_INSTS_VALU"] * clk),
                      "avg_waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / (1024 * n["SQ_INSTS_VALU"] * clk)})
            if "SQ_LDS_IDX_ACTIVE" in m:   # the LDS leg (tools/summarize_profile.py computes the same figures for the committed summaries)
                d.update({"lds_idx_active_frac_of_cu_cycles": m["SQ_LDS_IDX_ACTIVE"] / (256 * n["SQ_INSTS_VALU"] * clk),
                          "lds_bank_conflict_frac": m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m["SQ_LDS_IDX_ACTIVE"], 1),
                          "lds_wave_instructions_per_launch": m.get("SQ_INSTS_LDS"), "kernel_ns_in_sq_pass": n["SQ_INSTS_VALU"]})
        res[k] = d
    return res


def usable_cores():
    """host threads this process may actually run on: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def verify_against_oracle(res, cur, ref, w, h, sr, lq, bd, what="tables of the timed step", pred=None, sample=None):
    """res: int32 [2, n_refs, n_ctu, 593] (TComMv words, SADs) of the last timed step; reference 0 is checked bit-exactly
    against the oracle's exhaustive search on a sample of CTUs (default: corner, interior, partial bottom row).  pred: the per-CTU
    quarter-pel predictors the step searched around.  Raises SystemExit on a mismatch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from hmme import synth
    m = synth.MARGIN
    ctus_x, ctus_y = (w + 63) // 64, (h + 63) // 64
    if sample is None:   # the four corners, one CTU of each edge (the bottom row is partial at 2160 / 1080 rows), four interior CTUs
        sample = sorted({0, ctus_x - 1, ctus_x * (ctus_y - 1), ctus_x * ctus_y - 1,
                         ctus_x // 2, ctus_x * (ctus_y // 2), ctus_x * (ctus_y // 2 + 1) - 1, ctus_x * (ctus_y - 1) + ctus_x // 2,
                         ctus_x * (ctus_y // 2) + ctus_x // 3, ctus_x * (ctus_y // 3) + ctus_x // 2, ctus_x * (2 * ctus_y // 3) + 2 * ctus_x // 3,
                         ctus_x * (ctus_y // 4) + ctus_x // 5})
    mv = np.ascontiguousarray(res[0, 0]).view(np.int16).reshape(res.shape[2], 593, 2)
    sad = res[1, 0].view(np.uint32)
    t0 = time.time()
    for ctu in sample:
        ox, oy, osad = O.search_frame(cur, ref, (m, m), w, h, sr, pred, lq, 1, bd, ctu, 1, 1)
        if not (np.array_equal(mv[ctu, :, 0], ox[0]) and np.array_equal(mv[ctu, :, 1], oy[0]) and np.array_equal(sad[ctu], osad[0])):
            raise SystemExit(f"bench.py: {what} differ from the CPU oracle at CTU {ctu}: nothing reported")
    return {"ctus": [int(c) for c in sample], "slots": 593 * len(sample), "against": "oracle exhaustive search (bit-exact)", "seconds": round(time.time() - t0, 2)}


def cpu_baseline(cur, ref, w, h, sr, lq, bd=8, budget_s=14.0):
    """oracle legs on this node's host cores, bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from hmme import synth
    cores = min(usable_cores(), 64)
    m = synth.MARGIN
    ctus_x = (w + 63) // 64
    first = ctus_x * 3            # start on the 4th CTU row: full interior CTUs
    # exhaustive search (same arithmetic and same work as the GPU kernel)
    n = cores
    t0 = time.time(); O.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n, cores); dt = time.time() - t0
    n_full = max(cores, min(ctus_x * 16, int(n * (budget_s * 0.6) / max(dt, 1e-3)) // cores * cores))
    t0 = time.time(); O.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_full, cores); dt_full = time.time() - t0
    sads_full = n_full * 256 * (2 * sr + 1) ** 2
    # HM's default fast search (xTZSearch) over all 593 PU shapes of each CTU
    n_tz = cores * 4
    t0 = time.time(); O.tz_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_tz, cores, True); dt = time.time() - t0
    n_tz = max(cores, min(ctus_x * 28, int(n_tz * (budget_s * 0.3) / max(dt, 1e-3)) // cores * cores))
    probes = s4 = 0
    passes = 0
    t0 = time.time()
    while passes == 0 or (time.time() - t0 < 3.0 and passes < 200):   # repeat the sample until it is long enough to time
        p_, s_ = O.tz_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_tz, cores, True)
        probes += p_; s4 += s_; passes += 1
    dt_tz = time.time() - t0
    n_tz *= passes
    # the same exhaustive search on one core (SURVEY 8d asks for both figures)
    n_one = max(4, min(64, int(n_full / max(dt_full, 1e-3) / cores * 1.5)))
    t0 = time.time(); O.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_one, 1); dt_one = time.time() - t0
    # HM's OWN scalar xTZSearch (the reference's code compiled in place, oracle/_ref/libhmref.so -- prebuilt, it travels with the
    # snapshot; test infrastructure like the oracle) over the same CTUs on ONE core, as the single-threaded encoder runs it; its
    # tables must equal the oracle restatement's on the sample, or nothing is reported
    ref_tz = None
    try:
        have_ref = O.ref_available() and hasattr(O.ref(), "ref_tz_frame")
    except OSError:              # the prebuilt library does not load on this host: the oracle legs stand alone
        have_ref = False
    if have_ref:
        n_ref = 64
        t0 = time.time(); O.ref_tz_frame(cur, ref, (m, m), w, h, sr, LAMBDA, 1, bd, first, n_ref); dt = time.time() - t0
        n_ref = max(64, min(ctus_x * 28, int(n_ref * 3.0 / max(dt, 1e-3))))
        t0 = time.time(); rx, ry, rs = O.ref_tz_frame(cur, ref, (m, m), w, h, sr, LAMBDA, 1, bd, first, n_ref); dt_ref = time.time() - t0
        _, _, ox, oy, os_ = O.tz_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_ref, cores, True, True)
        if not (np.array_equal(rx, ox) and np.array_equal(ry, oy) and np.array_equal(rs, os_)):
            raise SystemExit("bench.py: the reference's xTZSearch and the oracle's restatement disagree on the timed sample: nothing reported")
        ref_tz = {"kind": "reference", "cores": 1, "ctus_per_s": round(n_ref / dt_ref, 1),
                  "sample": f"the reference's own TEncSearch::xTZSearch (oracle/_ref/libhmref.so), all 593 PUs of {n_ref} CTUs, 1 thread, {dt_ref:.2f} s; "
                            f"tables identical to the oracle restatement's on these CTUs"}
    # HM's OWN exhaustive search (TEncSearch::xPatternSearch, TEncSearch.cpp:3835-3897, the same libhmref.so): one PU at a time, all 593 PUs
    # of a few interior CTUs, one core -- the code the engine is bit-identical to, timed beside the port's one-core figure above; its
    # tables must equal the port's on these CTUs, or nothing is reported
    ref_full = None
    if have_ref:
        rx, ry, rs, dt_rf = O.ref_full_search_ctus(cur, ref, (m, m), w, h, sr, LAMBDA, 1, bd, first, min_ctus=4, max_ctus=16, budget_s=3.0)
        n_rf = rx.shape[0]
        ox, oy, os_ = O.search_frame(cur, ref, (m, m), w, h, sr, None, lq, 1, bd, first, n_rf, cores)
        if not (np.array_equal(rx, ox) and np.array_equal(ry, oy) and np.array_equal(rs, os_)):
            raise SystemExit("bench.py: the reference's xPatternSearch and the oracle's restatement disagree on the timed sample: nothing reported")
        ref_full = {"kind": "reference", "cores": 1, "value": round(n_rf * 256 * (2 * sr + 1) ** 2 / dt_rf / 1e9, 4), "unit": "GSAD/s",
                    "ctus_per_s": round(n_rf / dt_rf, 3),
                    "sample": f"the reference's own TEncSearch::xPatternSearch (oracle/_ref/libhmref.so) for each of the 593 PUs of {n_rf} interior CTUs "
                              f"of the same frame pair, 1 thread, {dt_rf:.2f} s; tables identical to the oracle restatement's on these CTUs",
                    "note": "HM searches every PU on its own (24 passes over the CTU's samples per candidate, FEN rows); the port forms all 593 sums "
                            "of a candidate from one pass, which is why its one-core figure is several times this one"}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {
        "value": round(sads_full / dt_full / 1e9, 4), "unit": "GSAD/s", "cores": cores, "kind": "port",
        "host": {"cpu": model, "logical_cpus": os.cpu_count(), "usable_by_this_process": usable_cores()},
        "one_core": {"value": round(n_one * 256 * (2 * sr + 1) ** 2 / dt_one / 1e9, 4), "unit": "GSAD/s", "ctus_per_s": round(n_one / dt_one, 2),
                     "sample": f"{n_one} CTUs, 1 thread, {dt_one:.2f} s"},
        "sample": f"oracle exhaustive search (xPatternSearch restatement, all 593 PUs) of {n_full} interior CTUs of the "
                  f"same frame pair, {cores} threads, {dt_full:.2f} s",
        "ctus_per_s": round(n_full / dt_full, 2),
        "tz": {"ctus_per_s": round(n_tz / dt_tz, 1), "gsad_equiv_per_s": round(s4 / dt_tz / 1e9, 4),
               "probes_per_s": round(probes / dt_tz, 0), "cores": cores,
               "sample": f"oracle xTZSearch restatement, all 593 PUs of {n_tz} CTUs, {dt_tz:.2f} s"},
        "tz_reference": ref_tz,
        "reference_full_search": ref_full,
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", default="2160p", help="one of %s, or WIDTHxHEIGHT (multiples of 8)" % ", ".join(sorted(SIZES)))
    ap.add_argument("--search-range", type=int, default=64)
    ap.add_argument("--bit-depth", type=int, default=8, help="8 = headline config; 10 + --search-range 128 = BASELINE config 5")
    ap.add_argument("--refs", type=int, default=1, help="reference pictures searched per step in one launch (lowdelay_P uses 4)")
    ap.add_argument("--pairs", type=int, default=1, help="DIFFERENT (current, reference) picture pairs searched per step in one launch "
                    "(hmme_search_pairs_device: what an open-loop pass over a sequence of small pictures does); excludes --refs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="headline only: no CPU legs, no oracle check, no extra configurations "
                    "(tools/profile_bench.sh profiles this command)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configurations and the refinement contents")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc passes of this command after the timed region "
                    "(the counters then come from the committed summary of the same library, if there is one)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu: rehearse the N > 1 code path on a one-GPU box (not a measurement)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--rank-timeout", type=float, default=600.0,
                    help="N > 1: seconds a rank may take from its start to the end of its first barrier (rendezvous, RCCL communicator) before it "
                         "gives up with exit code 3, naming itself and the stage it hung in; 0 = no limit")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="`python bench.py --gpus N` started without a launcher: seconds the parent lets the ranks it started run before it ends "
                         "their process group and exits 124")
    args = ap.parse_args(argv)
    if args.size not in SIZES:
        try:
            w, h = (int(v) for v in args.size.lower().split("x"))
        except ValueError:
            ap.error("--size: %s, or WIDTHxHEIGHT" % ", ".join(sorted(SIZES)))
        if w < 64 or h < 64 or w % 8 or h % 8:
            ap.error("--size: width and height must be multiples of 8, at least 64")
    if max(1, args.pairs) > 1 and args.refs > 1:
        ap.error("--pairs and --refs exclude each other")
    return args


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a child torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line and the exit code.  Nothing in THIS process has touched
    the GPU (torch is not even imported yet): a process that has initialised the GPU must not be replaced or forked on this pool.
    The child runs in a process group of its own; every rank leaves a file per start-up stage in a status directory.  When a rank's own
    watchdog ends it (exit code 3, --rank-timeout) torch.distributed.run takes the others down; should the launcher itself sit still past
    --launch-timeout, the parent ends the CHILD group (SIGTERM, then SIGKILL: the exact group it started, nothing matched by name),
    says which ranks never reached which stage, and exits 124."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    status = tempfile.mkdtemp(prefix="hmme_bench_status_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), HMME_BENCH_SELF_LAUNCHED="1",
               HMME_BENCH_STATUS_DIR=status)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("HMME_BENCH_TEST_LAUNCHER"):   # tests/test_shard_gloo.py: a stand-in for the launcher (a JSON argv) whose processes the parent must end
        cmd = json.loads(os.environ["HMME_BENCH_TEST_LAUNCHER"])
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    line = [None]

    def relay():
        for ln in child.stdout:                  # rank 0's line goes to stdout as the only line; everything else is diagnostics
            if ln.startswith('{"metric"'):
                line[0] = ln
            else:
                sys.stderr.write(ln)

    def end_child_group():
        """SIGTERM, then SIGKILL, to the session started above: exactly the launcher and the ranks it spawned, nothing matched by name"""
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(child.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue

    # the ranks live in a session of their own (so that the time-out below can end exactly them): a SIGTERM / SIGINT / SIGHUP that ends THIS
    # process -- a harness time-out, Ctrl-C -- would otherwise leave them on the GPUs with nobody reading their output
    class _Ended(BaseException):
        def __init__(self, signum):
            self.signum = signum

    def on_signal(signum, _frame):
        raise _Ended(signum)

    previous = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    timed_out = False
    rc = 1
    try:
        try:
            rc = child.wait(timeout=args.launch_timeout if args.launch_timeout > 0 else None)
        except subprocess.TimeoutExpired:
            timed_out = True
            reached = sorted(os.listdir(status))
            missing = [r for r in range(args.gpus) if f"rank{r}.first_barrier" not in reached]
            sys.stderr.write(f"bench.py: the {args.gpus} ranks did not finish within {args.launch_timeout:.0f} s; ranks that never passed their first barrier: "
                             f"{missing or 'none'}; stages reached: {reached}: ending the child process group {child.pid}\n")
            end_child_group()
            rc = 124
        reader.join(timeout=5.0)
        if rc != 0 and not timed_out:
            reached = sorted(os.listdir(status))
            missing = [r for r in range(args.gpus) if f"rank{r}.first_barrier" not in reached]
            sys.stderr.write(f"bench.py: the ranks exited with code {rc}; ranks that never passed their first barrier: {missing or 'none'}\n")
    except (_Ended, KeyboardInterrupt) as e:
        signum = getattr(e, "signum", signal.SIGINT)
        sys.stderr.write(f"bench.py: signal {signum} while the ranks were running: ending the child process group {child.pid}\n")
        rc = 128 + int(signum)
    finally:
        for sg in previous:                      # a second signal during the clean-up must not cut it short
            signal.signal(sg, signal.SIG_IGN)
        if child.poll() is None:
            end_child_group()
        shutil.rmtree(status, ignore_errors=True)
        for sg, h in previous.items():
            signal.signal(sg, h)
    if rc == 0 and line[0] is None:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    if line[0] is not None and rc == 0:
        sys.stdout.write(line[0])
        sys.stdout.flush()
    return rc


def device_identity(torch, local_rank):
    """what proves that a rank ran on a device of its own: name, PCI address and uuid of its GPU (from the runtime torch already holds:
    no second HIP library is loaded for this)"""
    props = torch.cuda.get_device_properties(local_rank)
    return {"name": props.name, "uuid": str(getattr(props, "uuid", "")), "cus": int(getattr(props, "multi_processor_count", 0)),
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}


def size_of(args_size):
    return SIZES[args_size] if args_size in SIZES else tuple(int(v) for v in args_size.lower().split("x"))


def time_search_config(torch, api, synth, eng, dev, w, h, bd, sr, steps, seed, label, pred=None):
    """one other configuration on this GPU, timed with HIP events on the launch stream and checked against the oracle;
    reported beside the headline, never part of `value`.  pred: int16 [n_ctu, 2] quarter-pel predictors -- every CTU's window is
    centred on its own predictor (xSetSearchRange, TEncSearch.cpp:3814-3830) and the MV cost is priced against it, as the encoder's
    caller does (TEncSearch.cpp:3732-3737); the job table is then rebuilt by every launch (a table without predictors is cached)"""
    cur, ref, _ = synth.make_pair(w, h, seed=seed, bit_depth=bd)
    m = synth.MARGIN
    pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
    pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
    n_ctu = api.load().hmme_num_ctus(w, h)
    fp = api.FrameParams(sr, 1, bd, 0, n_ctu)
    buf = torch.zeros((2, 1, n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev)
    d_pred = torch.from_numpy(np.ascontiguousarray(pred)).to(dev) if pred is not None else None
    stream = torch.cuda.current_stream().cuda_stream
    warm_clock(eng, (pc, pr), fp, buf, stream)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for i in range(steps + 2):
        if i == 2:
            ev[0].record()
        eng.search_pairs_device([pc], [pr], fp, d_pred.data_ptr() if d_pred is not None else None, buf[0].data_ptr(), buf[1].data_ptr(), stream)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    sads = work_4x4_sads(api, w, h, sr, pred)
    algo = algo_bytes_per_ctu(sr, bd) * n_ctu
    ctus_x, ctus_y = (w + 63) // 64, (h + 63) // 64
    # with predictors the windows of the picture's edge CTUs are the ones clipMv cuts: two corners, a left-edge, a right-edge, a top and a
    # (partial) bottom-row CTU and one interior CTU
    sample = None if pred is None else sorted({0, ctus_x - 1, ctus_x * (ctus_y // 2), ctus_x * (ctus_y // 2 + 1) - 1, ctus_x // 2,
                                                ctus_x * (ctus_y - 1) + ctus_x // 3, ctus_x * ctus_y - 1, ctus_x * (ctus_y // 3) + ctus_x // 2})
    out = {"workload": label, "steps": steps, "ms_per_step": round(ms, 4), "gsad_per_s": round(sads / (ms * 1e-3) / 1e9, 1),
           "ctus_per_s": round(n_ctu / (ms * 1e-3), 1), "dtype": "u8" if bd == 8 else "u16", "sads_4x4_per_frame": sads,
           "roofline": {"bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": algo},
           "verified": verify_against_oracle(buf.cpu().numpy(), cur, ref, w, h, sr, eng.lambda_q16, bd, what=f"tables of {label}", pred=pred, sample=sample)}
    if pred is not None:
        full = work_4x4_sads(api, w, h, sr)
        out["predictors"] = {"what": "synth.random_predictors(n_ctu, seed=4242, max_pel=16): seeded quarter-pel AMVP predictors, |component| <= 16 pel (SURVEY 8d)",
                             "work_vs_zero_predictors": round(sads / full, 4),
                             "work_is": "the candidates actually searched: each CTU's window after xSetSearchRange + clipMv around its predictor",
                             "job_table": "rebuilt by every launch (me_prep_jobs_kernel: device-side xSetSearchRange + clipMv); inside the timed region"}
    pc.close(); pr.close()
    return out


def time_sequence_config(torch, api, synth, eng, dev, refine=False):
    """BASELINE config 4 on THIS one GPU: 3840x2160, 64 pictures, the (current, reference) pairs of encoder_randomaccess_main.cfg
    (124 pairs), pictures streamed through a ring of plane slots while the GPU searches (hmme/sequence.py); wall clock around the
    whole pass, uploads included.  One CTU row of the first and the last pair is checked against the oracle.  refine: every pair's
    integer search is followed by the fractional refinement of its 593 x 2 040 winners on the same stream (what HM does per CTU:
    xMotionEstimation's search, TEncSearch.cpp:3749, then xPatternSearchFracDIF, :3798); refined slots of the same CTUs are checked too."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from hmme import sequence, shard
    w, h, n_frames, sr = 3840, 2160, 64, 64
    src = synth.Sequence(w, h, n_frames, seed=777, bit_depth=8)
    pairs = shard.gop_pairs(n_frames, "randomaccess")
    # four passes: the first includes allocations; the MEDIAN of the other three is the leg's figure, all three are listed (`seconds_passes`) and
    # the fastest is kept as `seconds_min` -- the pass is 0.3 s of a Python thread feeding three streams next to a reader thread, and one
    # pass in six came out 10 % long on an otherwise idle box (profiles/r05z_bench_default.json of build 123375ceac86b988: 0.344 s
    # against 0.312 s; the kernels were the same).  The tables checked below are the median pass's.
    runs = []
    for i in range(4):
        r = sequence.run_rank(eng, src, pairs, w, h, 8, sr, stream_mode=True, pairs_per_launch=1, device=dev, refine=refine)
        if i:
            runs.append(r)
            if len(runs) == 3:                   # keep the median pass's tables only (124 pairs of tables are 1.2 GB)
                runs.sort(key=lambda x: x["seconds"])
        r = None
    passes = [round(x["seconds"], 4) for x in runs]
    res = runs[1]
    runs = None
    dt = res["seconds"]
    n_ctu = api.load().hmme_num_ctus(w, h)
    ctus_x = (w + 63) // 64
    t0 = time.time()
    checked = []
    n_refined = 0
    for pi in (0, len(pairs) - 1):
        c, r = pairs[pi]
        first = ctus_x * 17
        ox, oy, osad = O.search_frame(src.padded(c), src.padded(r), (synth.MARGIN, synth.MARGIN), w, h, sr, None, eng.lambda_q16, 1, 8,
                                      first, 8, min(8, usable_cores()))
        mv = res["mv"][pi, first:first + 8].cpu().numpy()
        sad = res["sad"][pi, first:first + 8].cpu().numpy().view(np.uint32)
        if not (np.array_equal(mv[:, :, 0], ox) and np.array_equal(mv[:, :, 1], oy) and np.array_equal(sad, osad)):
            raise SystemExit(f"bench.py: config 4 tables of pair {pairs[pi]} differ from the CPU oracle: nothing reported")
        if refine:
            n_refined += check_refined_slots(O, src.padded(c), src.padded(r), w, h, 8, mv, res["qmv"][pi, first:first + 8].cpu().numpy(),
                                             res["cost"][pi, first:first + 8].cpu().numpy().view(np.uint32), range(first, first + 8, 3), eng.lambda_q16,
                                             f"config 4 with refinement, pair {pairs[pi]}", ctu_base=first)
        checked.append(list(pairs[pi]))
    sads = work_4x4_sads(api, w, h, sr)
    out = {"workload": "3840x2160 8-bit, 64 pictures, encoder_randomaccess_main GOP: 124 (current, reference) pairs, SearchRange=64, "
                       "streamed through ONE GPU (reader thread -> copy stream || compute stream)" +
                       (", every search followed by the fractional refinement (Hadamard) of its 593 x 2 040 winners" if refine else ""),
           "pairs": len(pairs), "seconds": round(dt, 4), "seconds_is": "the median of three timed passes", "seconds_passes": passes, "seconds_min": min(passes),
           "pairs_per_s": round(len(pairs) / dt, 1),
           "gsad_per_s": round(len(pairs) * sads / dt / 1e9, 1), "ctus_per_s": round(len(pairs) * n_ctu / dt, 1),
           "plane_slots": res["plane_slots"], "uploads": res["uploads"], "stages": res["stages"],
           "verified": {"pairs": checked, "ctus": [ctus_x * 17, 8], "slots": 2 * 8 * 593, "against": "oracle exhaustive search (bit-exact)",
                        "seconds": round(time.time() - t0, 2)}}
    if refine:
        out["verified"]["refined_slots"] = n_refined
        out["verified"]["against"] += "; refined slots against the oracle's xPatternSearchFracDIF restatement"
    return out


def time_sharded_sequence(torch, dist, api, synth, eng, dev, rank, world, w, h, bd, sr, backend, check_oracle, n_frames=64):
    """BASELINE config 4 as the job it names: the 124 (current, reference) pairs of a 64-picture random-access GOP
    (cfg/encoder_randomaccess_main.cfg:28-31) dealt pair p -> rank p mod N (hmme/shard.py), every rank streaming ITS pictures through a
    ring of plane slots while its GPU searches (hmme/sequence.py), the tables gathered to rank 0 in pair order with the one exchange step
    of the path (shard.gather_pair_results: grouped send / receive).  Every rank calls this; rank 0 returns the block for the line.
    Timed: barrier, every rank's pass over its share, the gather, barrier -- max over ranks; one untimed pass (allocations), then three
    timed ones, the median reported.  Strong scaling is read against the SAME job on rank 0 alone, timed in this run while the other ranks
    wait.  Each rank's CRC of its tables before the transfer must equal the CRC of what rank 0 holds for that rank afterwards, and eight
    CTUs of a pair the LAST rank searched are checked against the oracle."""
    from hmme import sequence, shard
    src = synth.Sequence(w, h, n_frames, seed=777, bit_depth=bd)
    pairs = shard.gop_pairs(n_frames, "randomaccess")
    counts = shard.pair_counts(len(pairs), world)
    on_dev = dev if backend == "nccl" else "cpu"
    n_ctu = api.load().hmme_num_ctus(w, h)

    # the one-GPU figure of this run: rank 0 searches all pairs alone (the others wait at the barrier below)
    # (plane slots, page-locked buffers and streams are allocated once and kept over all passes of this leg, as a long-running job keeps them;
    # both figures are wall clock around the whole pass)
    one = []
    with sequence.RankResources() as keep:
        if rank == 0:
            for i in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = sequence.run_rank(eng, src, pairs, w, h, bd, sr, stream_mode=True, pairs_per_launch=1, device=dev, resources=keep)
                torch.cuda.synchronize()
                if i:
                    one.append(time.perf_counter() - t0)
                r = None
        torch.cuda.synchronize()
        dist.barrier()
        job = shard.sharded_sequence_job(lambda share: sequence.run_rank(eng, src, share, w, h, bd, sr, stream_mode=True, pairs_per_launch=1, device=dev,
                                                                         resources=keep),
                                         pairs, passes=3, sync=torch.cuda.synchronize, reduce_device=on_dev)
    if rank != 0:
        return None
    crc_ok, per_rank, jobs, mv, sad = job["crc32_tables_match_per_rank"], job["per_rank"], job["seconds_passes"], job["mv"], job["sad"]
    if not all(crc_ok):
        raise SystemExit(f"bench.py: config 4 sharded: gathered tables differ from what the ranks computed (per-rank CRC match: {crc_ok}): nothing reported")
    dt = float(np.median(jobs))
    sads = work_4x4_sads(api, w, h, sr)
    out = {"workload": f"{w}x{h} {bd}-bit, {n_frames} pictures, encoder_randomaccess_main GOP: {len(pairs)} (current, reference) pairs, SearchRange={sr}, pair p -> rank p mod "
                       f"{world}; every rank streams its pictures (reader thread -> copy stream || compute stream); tables gathered to rank 0 in pair order",
           "pairs": len(pairs), "pair_counts": counts, "ideal_speedup_at_this_deal": round(len(pairs) / max(counts), 3),
           "seconds": round(dt, 4), "seconds_is": "barrier .. every rank's pass + gather .. barrier, max over ranks; the median of three timed passes",
           "seconds_passes": [round(v, 4) for v in jobs], "pairs_per_s": round(len(pairs) / dt, 1), "gsad_per_s": round(len(pairs) * sads / dt / 1e9, 1),
           "ctus_per_s": round(len(pairs) * n_ctu / dt, 1), "scaling": "strong",
           "one_gpu_same_run": {"what": "the same 124 pairs on rank 0 alone, streamed, the other ranks waiting", "seconds_passes": [round(v, 4) for v in one],
                                "pairs_per_s": round(len(pairs) / float(np.median(one)), 1)},
           "speedup_vs_one_gpu": round(float(np.median(one)) / dt, 3),
           "per_rank": per_rank, "crc32_tables_match_per_rank": crc_ok,
           "gathered_bytes": int(sum(counts[1:]) * n_ctu * api.NUM_PARTS * 8)}
    if check_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        p = min(world - 1, len(pairs) - 1)       # pair p was searched by rank p: the last rank's first pair
        c, r = pairs[p]
        ctus_x, ctus_y = (w + 63) // 64, (h + 63) // 64
        first, cnt = ctus_x * (ctus_y // 2), min(8, ctus_x)
        t0 = time.time()
        ox, oy, osad = O.search_frame(src.padded(c), src.padded(r), (synth.MARGIN, synth.MARGIN), w, h, sr, None, eng.lambda_q16, 1, bd, first, cnt,
                                      min(8, usable_cores()))
        gmv = mv[p, first:first + cnt].cpu().numpy()
        gsad = sad[p, first:first + cnt].cpu().numpy().view(np.uint32)
        if not (np.array_equal(gmv[:, :, 0], ox) and np.array_equal(gmv[:, :, 1], oy) and np.array_equal(gsad, osad)):
            raise SystemExit(f"bench.py: config 4 sharded: tables of pair {pairs[p]} (rank {p % world}) differ from the CPU oracle: nothing reported")
        out["verified"] = {"pair": list(pairs[p]), "pair_index": p, "searched_by_rank": p % world, "ctus": [first, cnt], "slots": cnt * 593,
                           "against": "oracle exhaustive search (bit-exact)", "seconds": round(time.time() - t0, 2)}
    return out


def check_refined_slots(O, cur, ref, w, h, bd, mv, qmv, cost, ctus, lq, what, ctu_base=0, n_random=20, seed=11):
    """refined tables (quarter-pel MV, cost) of the CTUs `ctus` against the oracle's xPatternSearchFracDIF restatement, on a seeded sample of
    slots per CTU plus the largest ones; mv / qmv / cost are indexed by (ctu - ctu_base).  Returns the number of slots checked."""
    from hmme import synth
    m = synth.MARGIN
    table = O.slot_table()
    ctus_x = (w + 63) // 64
    rs = np.random.default_rng(seed)
    n = 0
    for ctu in ctus:
        cx, cy = (ctu % ctus_x) * 64, (ctu // ctus_x) * 64
        k = ctu - ctu_base
        for s in list(rs.choice(593, size=n_random, replace=False)) + [592, 588, 0]:
            x, y, bw, bh = (int(v) for v in table[s])
            if cx + x + bw > w or cy + y + bh > h:
                continue   # slots beyond the picture edge: defined on the padding, not looked up by HM
            imv = (int(mv[k, s, 0]), int(mv[k, s, 1]))
            hx, hy, qx, qy, c = O.frac_refine(cur, (m + cx + x, m + cy + y), ref, (m + cx + x, m + cy + y), bw, bh, imv, (0, 0), lq, 1, bd)
            if (int(qmv[k, s, 0]), int(qmv[k, s, 1]), int(cost[k, s])) != (4 * imv[0] + 2 * hx + qx, 4 * imv[1] + 2 * hy + qy, c):
                raise SystemExit(f"bench.py: refinement of CTU {ctu} slot {s} ({what}) differs from the CPU oracle: nothing reported")
            n += 1
    return n


def time_refine_contents(torch, api, synth, eng, dev, w, h, bd, sr):
    """the refinement kernel's time depends on how much the 593 slots of a CTU share: one motion per 128x128 region (the bench
    content: the BEST case), 24-sample regions of their own motion under stronger noise, and unrelated pictures where nearly every
    slot has its own integer MV (the worst case).  Each is checked against the oracle on a sample of slots."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    m = synth.MARGIN
    n_ctu = api.load().hmme_num_ctus(w, h)
    fp = api.FrameParams(sr, 1, bd, 0, n_ctu)
    stream = torch.cuda.current_stream().cuda_stream
    d_mv = torch.zeros((n_ctu, api.NUM_PARTS, 2), dtype=torch.int16, device=dev)
    d_sad = torch.zeros((n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev)
    d_q, d_c = torch.zeros_like(d_mv), torch.zeros_like(d_sad)
    table = O.slot_table()
    ctus_x = (w + 63) // 64
    out = {}
    for content in ("coherent", "mixed", "noise"):
        if content == "coherent":
            cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd)
        elif content == "mixed":
            cur, ref, _ = synth.make_pair(w, h, seed=1234, bit_depth=bd, max_mv=10, region=24, noise_sigma=6.0)
        else:
            rng = np.random.default_rng(5)
            cur = synth.pad_plane(rng.integers(0, 1 << bd, size=(h, w)))
            ref = synth.pad_plane(rng.integers(0, 1 << bd, size=(h, w)))
        pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
        pc.upload_pel(cur, (m, m)); pr.upload_pel(ref, (m, m))
        warm_clock(eng, (pc, pr), fp, (d_mv, d_sad), stream)   # the GPU idled through the synthesis of this content
        eng.search_frame_device(pc, pr, fp, None, d_mv.data_ptr(), d_sad.data_ptr(), stream)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for i in range(7):
            if i == 2:
                ev[0].record()
            eng.refine_frame_multi_device(pc, [pr], fp, None, d_mv.data_ptr(), 1, d_q.data_ptr(), d_c.data_ptr(), stream)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        # what HM does per CTU, per picture pair here: the integer search, then the refinement of its winners, back to back on one stream
        for i in range(7):
            if i == 2:
                ev[0].record()
            eng.search_frame_device(pc, pr, fp, None, d_mv.data_ptr(), d_sad.data_ptr(), stream)
            eng.refine_frame_multi_device(pc, [pr], fp, None, d_mv.data_ptr(), 1, d_q.data_ptr(), d_c.data_ptr(), stream)
        ev[1].record()
        torch.cuda.synchronize()
        ms_both = ev[0].elapsed_time(ev[1]) / 5
        # the same pair four times in ONE launch (what a picture with four references asks for): the launch's tail -- it ends one job
        # time after its last job started -- is paid once
        mv4 = d_mv.unsqueeze(0).repeat(4, 1, 1, 1).contiguous()
        q4, c4 = torch.zeros_like(mv4), torch.zeros((4,) + tuple(d_c.shape), dtype=d_c.dtype, device=dev)
        for i in range(5):
            if i == 2:
                ev[0].record()
            eng.refine_frame_multi_device(pc, [pr] * 4, fp, None, mv4.data_ptr(), 1, q4.data_ptr(), c4.data_ptr(), stream)
        ev[1].record()
        torch.cuda.synchronize()
        ms4 = ev[0].elapsed_time(ev[1]) / 3
        if not (torch.equal(q4[3], d_q) and torch.equal(c4[3], d_c) and torch.equal(q4[0], d_q)):
            raise SystemExit(f"bench.py: refinement of four pairs in one launch differs from the single launches ({content} content): nothing reported")
        mv, qmv, cost = d_mv.cpu().numpy(), d_q.cpu().numpy(), d_c.cpu().numpy().view(np.uint32)
        n_checked = check_refined_slots(O, cur, ref, w, h, bd, mv, qmv, cost, (0, ctus_x * 16 + 29, n_ctu - 1), eng.lambda_q16, f"{content} content")
        # the integer tables the refinement started from are the back-to-back loop's: four CTUs of them against the oracle
        res = np.stack([d_mv.view(torch.int32).cpu().numpy().reshape(1, n_ctu, api.NUM_PARTS), d_sad.cpu().numpy().reshape(1, n_ctu, api.NUM_PARTS)])
        v_int = verify_against_oracle(res, cur, ref, w, h, sr, eng.lambda_q16, bd, what=f"integer tables of the search + refinement loop ({content} content)")
        out[content] = {"ms_per_step": round(ms, 4), "slots_per_s": round(n_ctu * api.NUM_PARTS / (ms * 1e-3)), "slots_verified": n_checked,
                        "ms_per_pair_at_four_pairs_per_launch": round(ms4 / 4, 4),
                        "search_plus_refine": {"ms_per_pair": round(ms_both, 4), "pairs_per_s": round(1e3 / ms_both, 1), "ctus_per_s": round(n_ctu / (ms_both * 1e-3), 1),
                                               "refine_share": round(ms / ms_both, 3),
                                               "verified": {"integer_ctus": v_int["ctus"], "integer_slots": v_int["slots"], "refined_slots": n_checked,
                                                            "against": "oracle exhaustive search + oracle xPatternSearchFracDIF restatement (bit-exact)"}}}
        pc.close(); pr.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # before torch is imported: this process never touches the GPU

    import zlib
    import torch
    import torch.distributed as dist
    from hmme import api, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                        # under a launcher the world it made is what runs
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: {torch.cuda.device_count()} visible "
                         f"(--share-gpu --backend gloo rehearses the N > 1 path on one GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HMME_BENCH_FORCE_DIST=1: run the collective path with world size 1 too (rehearses the RCCL calls on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("HMME_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    status_dir = os.environ.get("HMME_BENCH_STATUS_DIR")

    def mark(stage):                         # start-up stages of this rank, for the parent of a self-launched run
        if status_dir:
            try:
                open(os.path.join(status_dir, f"rank{rank}.{stage}"), "w").close()
            except OSError:
                pass

    watchdog = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # an N-rank run first happens unattended: a rank that cannot get through rendezvous and its first collective ends itself (exit 3)
        # with a line that names it and the stage, instead of leaving the job in a collective for the launcher's default half hour
        watchdog = shard.StartupWatchdog(rank, world, args.rank_timeout)
        mark("started")
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else "n/a (gloo)"
        except Exception as e:               # noqa: BLE001 -- a diagnostic must not end the run
            rccl = f"unknown ({e})"
        sys.stderr.write(f"bench.py: rank {rank}/{world} pid {os.getpid()} local_rank {local_rank}: hipGetDeviceCount {torch.cuda.device_count()}, "
                         f"device {device_identity(torch, local_rank)}, RCCL {rccl}, backend {args.backend}, "
                         f"MASTER {os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')}\n")
        sys.stderr.flush()
        watchdog.stage("init_process_group (rendezvous)")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        mark("process_group")
        watchdog.stage("rendezvous store count")
        seen = shard.rendezvous_report(rank, world, timeout_s=max(10.0, args.rank_timeout / 2) if args.rank_timeout > 0 else 120.0)
        if rank == 0 and seen is not None and seen != world:
            raise SystemExit(f"bench.py: only {seen} of {world} ranks reached the rendezvous store: nothing reported")
        watchdog.stage("first barrier (RCCL communicator)" if args.backend == "nccl" else "first barrier")
        stall = os.environ.get("HMME_BENCH_TEST_STALL", "")   # tests: "<rank>:<seconds>" -- that rank sits still in front of its first barrier
        if stall and int(stall.split(":")[0]) == rank:
            time.sleep(float(stall.split(":")[1]))
        dist.barrier()
        if args.backend == "nccl":
            torch.cuda.synchronize()
        watchdog.done()
        mark("first_barrier")

    w, h = size_of(args.size)
    sr = args.search_range
    bd = args.bit_depth
    eng = api.Engine(local_rank, 128)
    eng.set_lambda(LAMBDA)
    lq = eng.lambda_q16
    # frame shard: rank r searches frame pair r of the synthetic sequence (different seeds)
    cur, ref, _ = synth.make_pair(w, h, seed=1234 + rank, bit_depth=bd)
    pc, pr = eng.plane(w, h, bd), eng.plane(w, h, bd)
    pc.upload_pel(cur, (synth.MARGIN, synth.MARGIN))
    pr.upload_pel(ref, (synth.MARGIN, synth.MARGIN))
    n_pairs = max(1, args.pairs)
    n_refs = max(1, args.refs) if n_pairs == 1 else n_pairs      # searches per CTU position and launch
    ref_planes = [pr] + [eng.plane(w, h, bd) for _ in range(n_refs - 1)]
    cur_planes = [pc] * n_refs
    for i, pl in enumerate(ref_planes[1:]):
        c2, r2, _ = synth.make_pair(w, h, seed=5000 + 17 * i + rank, bit_depth=bd)
        pl.upload_pel(r2, (synth.MARGIN, synth.MARGIN))
        if n_pairs > 1:      # a current picture of its own
            cur_planes[i + 1] = eng.plane(w, h, bd)
            cur_planes[i + 1].upload_pel(c2, (synth.MARGIN, synth.MARGIN))
    n_ctu = api.load().hmme_num_ctus(w, h)
    fp = api.FrameParams(sr, 1, bd, 0, n_ctu)
    # one picture pair per rank and step.  Results of step k land in buffer k % 2 ([2, n_refs, n_ctu, 593] int32: TComMv
    # words and SADs) and travel to rank 0 asynchronously on RCCL's stream while step k+1 searches into the other
    # buffer (hmme/shard.py PipelinedGather; the same class runs under gloo in tests/test_shard_gloo.py)
    stream = torch.cuda.current_stream().cuda_stream
    pipe = shard.PipelinedGather(lambda: torch.zeros((2, n_refs, n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev),
                                 distributed=use_dist, async_op=(args.backend == "nccl"))
    events = []

    def launch(buf, k):
        ev = events[k - n_untimed] if k >= n_untimed else None
        if ev:
            ev[0].record()
        eng.search_pairs_device(cur_planes, ref_planes, fp, None, buf[0].data_ptr(), buf[1].data_ptr(), stream)
        if ev:
            ev[1].record()

    # untimed set-up: the power management takes ~10 launches (tens of ms) to bring an idle GPU to its sustained clock; without
    # this the first timed steps of a short run (small K and W) are measured at a lower clock than the rest (5 steps: +6 %)
    for _ in range(24):
        eng.search_pairs_device(cur_planes, ref_planes, fp, None, pipe.bufs[0][0].data_ptr(), pipe.bufs[0][1].data_ptr(), stream)
    torch.cuda.synchronize()
    n_untimed = args.warmup
    for _ in range(args.warmup):
        pipe.step(launch)
    pipe.drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    received0 = pipe.bytes_received
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.step(launch)
    pipe.drain()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0          # this rank's own K steps (before the closing barrier)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))   # HIP events on the launch stream

    # ---- evidence of the N-rank run (every rank contributes; rank 0 reports) ----------------------------------------------------
    multi = None
    if use_dist:
        last = pipe.last_local
        mine = {"rank": rank, "local_rank": local_rank, "device": device_identity(torch, local_rank), "kernel_ms": round(kernel_ms, 4),
                "step_ms": round(local_elapsed / args.steps * 1e3, 4), "pid": os.getpid(),
                "crc32_local_tables": zlib.crc32(last.cpu().numpy().tobytes())}    # this rank's tables of the last step, BEFORE any transfer
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        # the exchange step alone: three blocking gathers of one step's tables
        gt = []
        out_flat = (pipe.last_gathered.view((world * last.shape[0],) + tuple(last.shape[1:])) if rank == 0 and pipe.last_gathered is not None else None)
        keep = pipe.last_gathered.clone() if rank == 0 and pipe.last_gathered is not None else None
        got = 0
        for _ in range(3):
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            _, _, got = shard.gather_to_root(last, None, 0, out_flat)
            torch.cuda.synchronize()
            gt.append((time.perf_counter() - tg) * 1e3)
        # the same K steps with the exchange switched OFF (every rank; same buffers, same launches, no transfer): what a rank's step costs
        # when nothing travels.  step_ms - compute_only step_ms on rank 0 IS the price of receiving world - 1 tables per step beside a kernel
        # that fills the chip; on the other ranks it is the price of sending one.  Per-rank kernel times of this leg differ only by the
        # devices themselves (clocks, binning), per-rank (step - kernel) is host / launch time -- so a sub-linear curve can be read.
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        dist.barrier()
        tc = time.perf_counter()
        for k in range(args.steps):
            ev2[k][0].record()
            eng.search_pairs_device(cur_planes, ref_planes, fp, None, pipe.bufs[k & 1][0].data_ptr(), pipe.bufs[k & 1][1].data_ptr(), stream)
            ev2[k][1].record()
        torch.cuda.synchronize()
        co_local = time.perf_counter() - tc
        dist.barrier()
        co_elapsed = time.perf_counter() - tc
        tt = torch.tensor([co_elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        co_elapsed = float(tt.item())
        co_mine = {"kernel_ms": round(float(np.mean([a.elapsed_time(b) for a, b in ev2])), 4), "step_ms": round(co_local / args.steps * 1e3, 4)}
        co_all = [None] * world
        dist.all_gather_object(co_all, co_mine)
        sharded4 = None
        if not args.no_extras and n_refs == 1:   # BASELINE config 4 as a sharded job (every rank takes part)
            sharded4 = time_sharded_sequence(torch, dist, api, synth, eng, dev, rank, world, w, h, bd, sr, args.backend,
                                             check_oracle=not args.no_cpu_baseline)
        if rank == 0:
            distinct = sorted({e["device"]["pci_bus_id"] + "/" + e["device"]["uuid"] for e in everyone})
            if not args.share_gpu and len(distinct) != world:
                raise SystemExit(f"bench.py: {world} ranks ran on {len(distinct)} distinct devices ({distinct}): nothing reported")
            crc_ok = []
            for r in range(world):      # what rank 0 RECEIVED during the timed steps == what rank r computed
                blk = keep[r] if keep is not None else last
                crc_ok.append(zlib.crc32(blk.cpu().numpy().tobytes()) == everyone[r]["crc32_local_tables"])
            if not all(crc_ok):
                raise SystemExit(f"bench.py: gathered tables differ from what the ranks computed (per-rank CRC match: {crc_ok}): nothing reported")
            multi = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "self_launched": os.environ.get("HMME_BENCH_SELF_LAUNCHED") == "1",
                     "devices": [dict(e["device"], rank=e["rank"], local_rank=e["local_rank"], pid=e["pid"]) for e in everyone],
                     "distinct_devices": len(distinct), "shared_gpu_rehearsal": bool(args.share_gpu),
                     "per_rank_kernel_ms": [e["kernel_ms"] for e in everyone], "per_rank_step_ms": [e["step_ms"] for e in everyone],
                     "kernel_ms_min_max": [min(e["kernel_ms"] for e in everyone), max(e["kernel_ms"] for e in everyone)],
                     "step_ms_min_max": [min(e["step_ms"] for e in everyone), max(e["step_ms"] for e in everyone)],
                     "pairs_per_rank_per_step": [n_pairs] * world,
                     "gather": {"what": "every rank's tables of a step to rank 0 only (grouped point-to-point send / receive)",
                                "ms_blocking": round(float(np.median(gt)), 4), "bytes_received_by_rank0_per_step": int(got),
                                "bytes_received_in_timed_steps": int(pipe.bytes_received - received0), "overlapped_with_next_search": args.backend == "nccl"},
                     "crc32_tables_match_per_rank": crc_ok}
            co_step, co_kern = [e["step_ms"] for e in co_all], [e["kernel_ms"] for e in co_all]
            multi["compute_only"] = {
                "what": "the same K steps on every rank with the exchange switched off (no gather): barrier + synchronize on both sides, max over ranks",
                "ms_per_step": round(co_elapsed / args.steps * 1e3, 4),
                "gsad_per_s": round(work_4x4_sads(api, w, h, sr) * n_refs * world * args.steps / co_elapsed / 1e9, 2),
                "per_rank_step_ms": co_step, "per_rank_kernel_ms": co_kern,
                "step_ms_min_max": [min(co_step), max(co_step)], "kernel_ms_min_max": [min(co_kern), max(co_kern)],
                "per_rank_host_ms": [round(s_ - k_, 4) for s_, k_ in zip(co_step, co_kern)]}
            multi["exchange_cost"] = {
                "what": "step time with the pipelined gather minus step time without it, rank by rank: on rank 0 the price of receiving world - 1 tables "
                        "per step while its own search kernel fills the chip, elsewhere the price of sending one; kernel_ms_delta is the part of it the "
                        "search kernel itself ran longer (CU slots / HBM shared with the transfer), the rest is host and stream time",
                "per_rank_step_ms_delta": [round(e["step_ms"] - c, 4) for e, c in zip(everyone, co_step)],
                "per_rank_kernel_ms_delta": [round(e["kernel_ms"] - c, 4) for e, c in zip(everyone, co_kern)],
                "job_ms_per_step_delta": round(elapsed / args.steps * 1e3 - co_elapsed / args.steps * 1e3, 4),
                "frac_of_step": round(1.0 - co_elapsed / elapsed, 4)}
            if world > 1 and not args.no_cpu_baseline:   # one non-zero rank's tables against the oracle (its frame pair is regenerated here)
                r = world - 1
                c_r, r_r, _ = synth.make_pair(w, h, seed=1234 + r, bit_depth=bd)
                multi["verified_rank"] = dict(verify_against_oracle(keep[r].cpu().numpy(), c_r, r_r, w, h, sr, lq, bd, what=f"tables gathered from rank {r}"), rank=r)

    if rank == 0:
        sads = work_4x4_sads(api, w, h, sr) * n_refs
        total_sads = sads * world * args.steps
        algo_bytes = algo_bytes_per_ctu(sr, bd) * n_ctu * n_refs
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        prof = pmc_profile(args.size, sr, bd) if n_refs == 1 else {}
        search_kernel = "me_search_kernel" if bd == 8 else "me_search16_kernel"
        kprof = prof.get("kernels", {}).get(search_kernel, {})
        # counters of THIS run where the profiler is at hand (N = 1, the full default-style run): child rocprofv3 passes of this very
        # command; otherwise the committed summary taken on the library with the same build id
        live, live_all = {}, {}
        if world == 1 and n_refs == 1 and not args.no_cpu_baseline and not args.no_live_pmc:
            passthrough = ["--size", args.size, "--search-range", str(sr), "--bit-depth", str(bd)]
            live_all = live_pmc(passthrough, [search_kernel, "me_frac_kernel"])
            live = live_all.get(search_kernel, {})
            if live.get("hbm_traffic_bytes_per_launch") is not None:
                kprof = dict(kprof, **live)
        same_run = live.get("hbm_traffic_bytes_per_launch") is not None
        traffic = kprof.get("hbm_traffic_bytes_per_launch")
        out = {
            "metric": "GSAD/s", "value": round(total_sads / elapsed / 1e9, 2), "unit": "GSAD/s (4x4-block SAD evaluations)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8" if bd == 8 else "u16", "data": "synthetic",
            "ctus_per_s": round(n_ctu * n_refs * world * args.steps / elapsed, 1),
            "config": {"workload": f"{w}x{h} {bd}-bit luma, lowdelay_P_main{'' if bd == 8 else '10'} (FEN=1), SearchRange={sr}, CTU=64, exhaustive "
                                   f"integer search of all 593 PU shapes, " + (f"{n_pairs} picture pairs" if n_pairs > 1 else f"{n_refs} reference picture{'s' if n_refs > 1 else ''}") + f" per launch, {n_ctu} CTUs per frame",
                       "frames_per_step": world * n_pairs, "parallelism": f"frame-shard x{world}", "lambda": LAMBDA,
                       "collective": f"gather to rank 0: grouped send/recv ({args.backend}), world {world}" if use_dist else "none (one rank)",
                       "sads_4x4_per_frame": sads},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": int(traffic) if traffic is not None else None,
                         # counters: separate rocprofv3 --pmc passes of this command -- started by this very run after its timed region
                         # (true), or taken earlier on the library with the same build id and committed under profiles/ (false)
                         "traffic_same_run": (same_run if traffic is not None else None),
                         "traffic_source": (("rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE: one pass each) of `bench.py --steps 3 --no-cpu-baseline` "
                                             "started by this run, %.0f s" % live["seconds"]) if same_run else
                                            ("profiles/latest_pmc_%s.json" % profile_label(args.size, sr, bd) if traffic is not None else None)),
                         "achieved_is": "ALGORITHMIC bytes per launch (SURVEY 8d: CTU + window + results per CTU-search) / kernel time; "
                                        "`traffic` is the MEASURED HBM bytes per launch (neighbouring windows hit the XCD's L2), "
                                        "`traffic_gbs` = traffic / kernel time",
                         "traffic_gbs": round(traffic / (kernel_ms * 1e-3) / 1e9, 2) if traffic is not None else None,
                         "kernel": ("me_search_kernel<1, 0> (current picture read from the plane's CTU-blocked copy: one instantiation for every picture size)" if bd == 8
                                    else "me_search16_kernel<1, PDW> (+ merge-table preset and finalize)"),
                         "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "kernel is VALU-bound (1664 abs-diff ops per unique input byte at SR 64, SURVEY 8d); "
                                 "see valu_roofline and DESIGN.md 5 for the instruction-issue roofline"},
        }
        if multi is not None:
            out["multi_gpu"] = multi
            if sharded4 is not None:
                out["configs"] = {"config4_sharded": sharded4}
        if prof.get("stale"):
            out["roofline"]["traffic_note"] = "profiles/latest_pmc_* was taken on a library with another build id than the one loaded: counters withheld"
        if kprof.get("valu_busy_frac") is not None:   # what actually binds: integer VALU issue
            # abs-diff operations the kernel performs: every candidate touches each of the CTU's 4096 samples once (FEN halves
            # nothing in the kernel: the even-row sums are the first half of the full sums)
            absdiff = sads * 16 / (kernel_ms * 1e-3)
            # measured SAD-only ceiling (profiles/archive/r01_valu_lds_rates_ubench.txt, 2 waves/SIMD): v_qsad_pk_u16_u8 issues 0.55
            # wave-instructions per ns and CU, 1024 abs-diffs each -> 144e12 abs-diff/s on 256 CUs; v_sad_u16 1.917 per ns and CU,
            # 128 abs-diffs each -> 62.8e12
            ceiling = 144.2e12 if bd == 8 else 62.8e12
            out["valu_roofline"] = {"bound": "valu-issue", "same_run": live.get("valu_busy_frac") is not None, "valu_busy_frac": round(kprof["valu_busy_frac"], 4),
                                    "valu_wave_instructions_per_launch": int(kprof["valu_wave_instructions_per_launch"]),
                                    "effective_clock_ghz": round(kprof.get("effective_clock_ghz", 0.0), 3),
                                    "abs_diff_per_s": round(absdiff, 0),
                                    "sad_only_ceiling_abs_diff_per_s": ceiling,
                                    "frac_of_sad_only_ceiling": round(absdiff / ceiling, 4),
                                    "frac_of_nominal_sad_u8_peak": round(absdiff / 314.6e12, 4),
                                    "note": "sad_only_ceiling = the measured issue rate of the leaf instruction alone (v_qsad_pk_u16_u8 / v_sad_u16 "
                                            "micro-benchmarks under profiles/), i.e. a kernel whose reduction tree and arg-min cost nothing; it is "
                                            "0.46 (u8) of SURVEY 8d's paper figure 314.6e12 (v_sad_u8 at 2 cycles per wave-instruction)",
                                    "source": ("rocprofv3 --pmc passes of this command started by this run (library build id %s)" % library_build_id())
                                              if live.get("valu_busy_frac") is not None else
                                              "profiles/latest_pmc_%s.json (rocprofv3 --pmc passes of this command on the library with the "
                                              "same build id: %s)" % (profile_label(args.size, sr, bd), library_build_id())}
            if kprof.get("avg_waves_per_simd") is not None:
                out["valu_roofline"]["avg_waves_per_simd"] = round(kprof["avg_waves_per_simd"], 3)
            # inside `roofline` as well, so that the parsed block alone says what binds: `bound` / `frac` stay the contract's HBM figure
            out["roofline"]["binding"] = "valu-issue"
            out["roofline"]["valu_busy_frac"] = out["valu_roofline"]["valu_busy_frac"]
            out["roofline"]["frac_of_sad_only_ceiling"] = out["valu_roofline"]["frac_of_sad_only_ceiling"]
            if kprof.get("lds_idx_active_frac_of_cu_cycles") is not None and kprof.get("lds_wave_instructions_per_launch"):
                # SURVEY 8d's LDS leg: bytes the kernel's LDS reads deliver per second against the chip's ~150 TB/s (256 CUs x 128 B/clk x
                # ~2.4 GHz x 2 ports, MI355X_MICROARCH.md).  Nearly every LDS instruction of the kernel is the ds_read2_b32 that feeds one
                # v_qsad_pk_u16_u8: 8 window bytes per lane
                lds_bytes = kprof["lds_wave_instructions_per_launch"] * 64 * 8
                out["roofline"]["lds"] = {"lds_idx_active_frac_of_cu_cycles": round(kprof["lds_idx_active_frac_of_cu_cycles"], 4),
                                          "lds_bank_conflict_frac": round(kprof.get("lds_bank_conflict_frac", 0.0), 4),
                                          "lds_wave_instructions_per_launch": int(kprof["lds_wave_instructions_per_launch"]),
                                          "tb_per_s": round(lds_bytes / (kernel_ms * 1e-3) / 1e12, 2), "peak_tb_per_s": 150.0,
                                          "frac_of_peak": round(lds_bytes / (kernel_ms * 1e-3) / 1e12 / 150.0, 4)}
        # the step after the path, reported beside the headline, never part of `value`
        d_q = torch.zeros((n_refs, n_ctu, api.NUM_PARTS, 2), dtype=torch.int16, device=dev)
        d_c = torch.zeros((n_refs, n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev)
        last = pipe.last_local
        d_mv16 = last[0].view(torch.int16).contiguous()   # TComMv words -> int16 (x, y) pairs
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        scratch_tables = torch.zeros((2, n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev)
        warm_clock(eng, (pc, pr), fp, scratch_tables, stream, launches=12)   # this process left the GPU to the counter passes above
        for i in range(7):
            if i == 2:
                ev[0].record()
            eng.refine_pairs_device(cur_planes, ref_planes, fp, None, d_mv16.data_ptr(), 1, d_q.data_ptr(), d_c.data_ptr(), stream)
        ev[1].record()
        torch.cuda.synchronize()
        r_ms = ev[0].elapsed_time(ev[1]) / 5
        out["refine"] = {"what": "xPatternSearchFracDIF (half + quarter-pel, Hadamard) for all 593 slots of every CTU, on the integer winners",
                         "kernel": "me_frac_kernel<1, %d>" % (1 if bd == 8 else 2), "ms_per_step": round(r_ms, 4),
                         "slots_per_s": round(n_ctu * n_refs * api.NUM_PARTS / (r_ms * 1e-3)),
                         "content": "the bench pictures: one motion per 128x128 region, i.e. the refinement kernel's BEST case "
                                    "(slots of a CTU share their patches); see by_content for the others"}
        extras = n_refs == 1 and n_pairs == 1 and world == 1 and not args.no_cpu_baseline
        # beside the headline, never `value`: the same picture against the FOUR reference pictures a lowdelay_P_main picture searches
        # (cfg/encoder_lowdelay_P_main.cfg:24-27) in one launch -- what the encoder configuration the metric is quoted on asks of a step
        # (not under --no-cpu-baseline: tools/profile_bench.sh profiles that command, and launches of another size would enter the
        # per-kernel averages of its rocprofv3 summaries)
        if extras:
            refs4 = [pr] + [eng.plane(w, h, bd) for _ in range(3)]
            for i, pl in enumerate(refs4[1:]):
                _, r2, _ = synth.make_pair(w, h, seed=7000 + 13 * i, bit_depth=bd)
                pl.upload_pel(r2, (synth.MARGIN, synth.MARGIN))
            b4 = torch.zeros((2, 4, n_ctu, api.NUM_PARTS), dtype=torch.int32, device=dev)
            reps4 = max(3, min(10, args.steps // 2))
            for i in range(reps4 + 2):
                if i == 2:
                    ev[0].record()
                eng.search_frame_multi_device(pc, refs4, fp, None, b4[0].data_ptr(), b4[1].data_ptr(), stream)
            ev[1].record()
            torch.cuda.synchronize()
            ms4 = ev[0].elapsed_time(ev[1]) / reps4
            out["four_references_per_launch"] = {"what": "one current picture against 4 reference pictures in ONE launch (hmme_search_frame_multi_device)",
                                                 "ms_per_launch": round(ms4, 4), "gsad_per_s": round(4 * sads / (ms4 * 1e-3) / 1e9, 1),
                                                 "ctus_per_s": round(4 * n_ctu / (ms4 * 1e-3), 1)}
            for pl in refs4[1:]:
                pl.close()
        fprof = live_all.get("me_frac_kernel") or prof.get("kernels", {}).get("me_frac_kernel", {})
        if fprof.get("hbm_traffic_bytes_per_launch") is not None:   # counters of the refinement launches of the same child runs as the search kernel's
            out["refine"]["counters_same_run"] = "me_frac_kernel" in live_all
            out["refine"]["hbm_traffic_bytes_per_launch"] = int(fprof["hbm_traffic_bytes_per_launch"])
            if fprof.get("hbm_write_bytes_per_launch") is not None:
                out["refine"]["hbm_write_bytes_per_launch"] = int(fprof["hbm_write_bytes_per_launch"])
                out["refine"]["result_bytes_per_launch"] = n_ctu * n_refs * api.NUM_PARTS * 8
            if fprof.get("scratch_bytes_per_lane") is not None:
                out["refine"]["scratch_bytes_per_lane"] = int(fprof["scratch_bytes_per_lane"])
            out["refine"]["valu_busy_frac"] = round(fprof.get("valu_busy_frac", 0.0), 4)
            if fprof.get("avg_waves_per_simd") is not None:
                out["refine"]["avg_waves_per_simd"] = round(fprof["avg_waves_per_simd"], 3)
            if fprof.get("lds_bank_conflict_frac") is not None:
                out["refine"]["lds_bank_conflict_frac"] = round(fprof["lds_bank_conflict_frac"], 4)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks would wait on it
            # what was timed is what is reported: tables of the LAST timed step against the CPU oracle on a CTU sample
            # (corner, interior, partial bottom row); a mismatch fails the run
            res = last.cpu().numpy()
            out["verified"] = verify_against_oracle(res, cur, ref, w, h, sr, lq, bd)
        if extras and not args.no_extras and (w, h, bd, sr) == (3840, 2160, 8, 64):
            # the other single-GPU BASELINE configurations and the refinement kernel's content dependence, each timed with HIP events
            # and checked against the oracle in this same process (BASELINE.json configs 2, 5 and 4-on-one-GPU)
            t_x = time.time()
            out["refine"]["by_content"] = time_refine_contents(torch, api, synth, eng, dev, w, h, bd, sr)
            out.setdefault("configs", {}).update({
                "config2_1080p_sr64": time_search_config(torch, api, synth, eng, dev, 1920, 1080, 8, 64, 20, 1234,
                                                         "1920x1080 8-bit, SearchRange=64 (BASELINE config 2), one reference per launch"),
                "config5_2160p_10bit_sr128": time_search_config(torch, api, synth, eng, dev, 3840, 2160, 10, 128, 5, 1234,
                                                                "3840x2160 10-bit, SearchRange=128, packed-u16 SAD path (BASELINE config 5)"),
                # SURVEY 8d's second measurement: the headline workload with every CTU's window centred on a seeded predictor of up
                # to +-16 pel, as the encoder's caller centres it on the AMVP predictor
                "with_random_predictors": time_search_config(torch, api, synth, eng, dev, 3840, 2160, 8, 64, 10, 1234,
                                                             "3840x2160 8-bit, SearchRange=64, per-CTU random predictors <= +-16 pel (SURVEY 8d)",
                                                             pred=synth.random_predictors(n_ctu, seed=4242, max_pel=16)),
                # pictures that do not fill whole rounds of the chip's 512 search workgroups, one pair per launch: 720p is 240 jobs (all "tail": equal
                # segments of the jobs' task list), 1440p 512 whole jobs + 408 as segments in the same launch (DESIGN.md 4.1 "Tails")
                "small_pictures_one_pair_per_launch": {
                    "1280x720": time_search_config(torch, api, synth, eng, dev, 1280, 720, 8, 64, 20, 1234, "1280x720 8-bit, SearchRange=64, one pair per launch (240 CTU searches on 512 workgroup slots)"),
                    "2560x1440": time_search_config(torch, api, synth, eng, dev, 2560, 1440, 8, 64, 20, 1234, "2560x1440 8-bit, SearchRange=64, one pair per launch (920 CTU searches = 512 + 408)")},
                "config4_2160p_randomaccess_64_pictures_one_gpu": time_sequence_config(torch, api, synth, eng, dev),
                "config4_with_refinement": time_sequence_config(torch, api, synth, eng, dev, refine=True),
            })
            out["configs"]["seconds"] = round(time.time() - t_x, 1)
            # the end-to-end figure an encoder integrator needs: integer search + fractional refinement of a 2160p pair, per content
            out["search_plus_refine"] = {"what": "hmme_search_frame_device then hmme_refine_frame_multi_device (Hadamard) of the same 2160p pair, back to back on one stream "
                                                 "(TEncSearch.cpp:3749 then :3798 for every PU of every CTU)",
                                         "by_content": {k: v["search_plus_refine"] for k, v in out["refine"]["by_content"].items()}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cur, ref, w, h, sr, lq, bd)
        print(json.dumps(out), flush=True)
    for pl in set(cur_planes) | set(ref_planes):
        pl.close()
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
