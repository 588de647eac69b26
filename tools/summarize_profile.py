#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run into the two files committed under profiles/:
   <tag>_kernel_stats_2160p_sr64.csv   rocprofv3 --kernel-trace --stats summary
   <tag>_pmc_summary_2160p_sr64.json   per-dispatch means of every counter of me_search_kernel + derived figures
and refresh profiles/latest_pmc_2160p_sr64.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
stats = glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(prof, f"{tag}_kernel_stats_2160p_sr64.csv"))
if os.path.exists(os.path.join(out_dir, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(out_dir, "bench_under_rocprof.json"), os.path.join(prof, f"{tag}_bench_under_rocprof.json"))
summary = {"passes": {}}
counters = {}
for path in sorted(glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc, dur, meta = collections.defaultdict(list), [], {}
    for r in csv.DictReader(open(path)):
        if "me_search" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
    if not acc:
        continue
    name = os.path.basename(os.path.dirname(os.path.dirname(path)))
    means = {k: sum(v) / len(v) for k, v in acc.items()}
    counters.update(means)
    summary["passes"][name] = {"per_dispatch_mean": means, "dispatches": len(next(iter(acc.values()))),
                               "mean_kernel_ns_in_this_pass": sum(dur) / len(dur), "dispatch": meta}
d = {}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = counters["FETCH_SIZE"], counters["WRITE_SIZE"]
    d.update({"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_read_bytes_corrected_x2_gfx950": f * 1024 * 2,
              "hbm_write_bytes": w * 1024, "hbm_traffic_bytes_per_launch": f * 1024 * 2 + w * 1024,
              "hbm_note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of a coalesced stream -> doubled; "
                          "WRITE_SIZE as is; each counter from its own --pmc pass"})
if "SQ_INSTS_VALU" in counters:
    kns = summary["passes"][[k for k in summary["passes"] if "SQ_WAVES" in k][0]]["mean_kernel_ns_in_this_pass"]
    d.update({"valu_wave_instructions_per_launch": counters["SQ_INSTS_VALU"],
              "valu_instructions_per_wave": counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"],
              "valu_active_cycles_x4": counters["SQ_ACTIVE_INST_VALU"] * 4,
              "lds_bank_conflict_frac": counters["SQ_LDS_BANK_CONFLICT"] / max(counters["SQ_LDS_IDX_ACTIVE"], 1),
              "kernel_ns_in_sq_pass": kns})
    if "GRBM_GUI_ACTIVE" in counters:
        kns_g = summary["passes"][[k for k in summary["passes"] if "GRBM" in k][0]]["mean_kernel_ns_in_this_pass"]
        clk = counters["GRBM_GUI_ACTIVE"] / 8 / kns_g   # GHz (sum over 8 XCDs)
        d["effective_clock_ghz"] = clk
        d["valu_busy_frac"] = counters["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * kns * clk)   # 1024 SIMDs
summary["derived"] = d
p = os.path.join(prof, f"{tag}_pmc_summary_2160p_sr64.json")
json.dump(summary, open(p, "w"), indent=1)
shutil.copy(p, os.path.join(prof, "latest_pmc_2160p_sr64.json"))
print(json.dumps(d, indent=1))
