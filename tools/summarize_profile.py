#!/usr/bin/env python3
"""Condense one configuration of a tools/profile_bench.sh run into the files committed under profiles/:
   <tag>_kernel_stats_<label>.csv   rocprofv3 --kernel-trace --stats summary
   <tag>_pmc_summary_<label>.json   per-dispatch means of every counter, per engine kernel, + derived figures
and refresh profiles/latest_pmc_<label>.json (read by bench.py for roofline.traffic / valu_roofline; it carries the hash of
the loaded library's hmme_build_id() so that bench.py can tell whether the counters belong to the library it runs).
usage: summarize_profile.py <out_dir> <tag> <label>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out_dir, tag, label = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
KERNELS = ("me_search_kernel", "me_search16_kernel", "me_frac_kernel")

stats = glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(prof, f"{tag}_kernel_stats_{label}.csv"))
if os.path.exists(os.path.join(out_dir, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(out_dir, "bench_under_rocprof.json"), os.path.join(prof, f"{tag}_bench_under_rocprof_{label}.json"))


def kernel_of(name):
    for k in KERNELS:
        if name.startswith(k + "<") or name.startswith("void hmme::" + k + "<") or ("::" + k + "<") in name or name.startswith(k + "("):
            return k
    return None


summary = {"label": label, "passes": {}, "kernels": {}}
counters = collections.defaultdict(dict)     # kernel -> counter -> per-dispatch mean
pass_ns = collections.defaultdict(dict)      # kernel -> pass name -> mean kernel ns in that pass
for path in sorted(glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    name = os.path.basename(path[:path.index(os.sep, path.index("pmc_"))]) if os.sep in path[path.index("pmc_"):] else "pmc"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)      # kernel -> dispatch id -> ns
    meta = {}
    for r in csv.DictReader(open(path)):
        k = kernel_of(r["Kernel_Name"])
        if k is None:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        meta[k] = {f: r[f] for f in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if f in r}
    for k, cs in acc.items():
        means = {c: sum(v) / len(v) for c, v in cs.items()}
        counters[k].update(means)
        ns = sum(dur[k].values()) / len(dur[k])
        pass_ns[k][name] = ns
        summary["passes"].setdefault(name, {})[k] = {"per_dispatch_mean": means, "dispatches": len(dur[k]), "mean_kernel_ns_in_this_pass": ns,
                                                     "dispatch": meta[k]}

for k, c in counters.items():
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = c["FETCH_SIZE"], c["WRITE_SIZE"]
        d.update({"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_read_bytes_corrected_x2_gfx950": f * 1024 * 2,
                  "hbm_write_bytes": w * 1024, "hbm_traffic_bytes_per_launch": f * 1024 * 2 + w * 1024,
                  "hbm_note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of a coalesced stream -> doubled; "
                              "WRITE_SIZE as is; each counter from its own --pmc pass"})
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        kns = pass_ns[k].get("pmc_SQ_WAVES")
        d.update({"valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                  "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1),
                  "lds_wave_instructions_per_launch": c.get("SQ_INSTS_LDS"),
                  "valu_active_cycles_x4": c["SQ_ACTIVE_INST_VALU"] * 4,
                  "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1),
                  "kernel_ns_in_sq_pass": kns})
        if "GRBM_GUI_ACTIVE" in c and kns:
            kns_g = pass_ns[k].get("pmc_GRBM_GUI_ACTIVE", kns)
            clk = c["GRBM_GUI_ACTIVE"] / 8 / kns_g   # GHz (the counter sums over the 8 XCDs)
            d["effective_clock_ghz"] = clk
            d["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * kns * clk)   # 1024 SIMDs
            if "SQ_LDS_IDX_ACTIVE" in c:
                d["lds_idx_active_frac_of_cu_cycles"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * kns * clk)
            if "SQ_ACTIVE_INST_LDS" in c:
                kns2 = pass_ns[k].get("pmc_SQ_INSTS_SALU", kns)
                d["lds_inst_active_frac"] = c["SQ_ACTIVE_INST_LDS"] * 4 / (1024 * kns2 * clk)
            if "SQ_WAIT_INST_LDS" in c and "SQ_WAVE_CYCLES" in c:
                d["wait_inst_lds_per_wave_cycle"] = c["SQ_WAIT_INST_LDS"] / max(c["SQ_WAVE_CYCLES"], 1)
            if "SQ_WAVE_CYCLES" in c:   # how a wave spends its life (SQ_WAVE_CYCLES counts quad-cycles of resident waves) and how many are resident
                d["avg_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (1024 * kns * clk)
                for src, dst in (("SQ_WAIT_ANY", "wait_any_per_wave_cycle"), ("SQ_WAIT_INST_ANY", "wait_inst_any_per_wave_cycle"),
                                 ("SQ_ACTIVE_INST_ANY", "active_inst_any_per_wave_cycle")):
                    if src in c:
                        d[dst] = c[src] / max(c["SQ_WAVE_CYCLES"], 1)
    summary["kernels"][k] = d

try:
    import bench
    summary["library_build_id"] = bench.library_build_id()
except Exception as e:   # noqa: BLE001 -- the summary is still useful without the tie to the library
    summary["library_build_id"] = None
    summary["hash_error"] = repr(e)
p = os.path.join(prof, f"{tag}_pmc_summary_{label}.json")
json.dump(summary, open(p, "w"), indent=1)
if not label.startswith("quick_"):   # the tooling's self-test does not replace the counters bench.py quotes
    shutil.copy(p, os.path.join(prof, f"latest_pmc_{label}.json"))
print(json.dumps(summary["kernels"], indent=1))
