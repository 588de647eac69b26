#!/bin/bash
# counters of me_frac_kernel on one content: bash tools/frac_counters.sh <content> <outdir>   (rocprofv3 --pmc passes of tools/refine_rate.py)
C=${1:-noise}; OUT=${2:-gpurun_out/frac_counters}; mkdir -p $OUT; export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  p=$(echo $pass | cut -d' ' -f1)
  WARM=20 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${C}_$p -- python3 tools/refine_rate.py 3840x2160 8 $C > /dev/null 2> $OUT/${C}_$p.err || echo "pass $p failed"
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for path in glob.glob("$OUT/${C}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "me_frac_kernel<1" in r["Kernel_Name"]:     # the Hadamard launches
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); dur[r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}; n = {k: sum(v) / len(v) for k, v in dur.items()}
clk = m["GRBM_GUI_ACTIVE"] / 8 / n["GRBM_GUI_ACTIVE"]
d = {"content": "$C", "kernel_us": round(n["SQ_INSTS_VALU"] / 1e3, 1), "clock_ghz": round(clk, 3),
     "valu_busy_frac": round(m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * n["SQ_INSTS_VALU"] * clk), 4),
     "avg_waves_per_simd": round(m["SQ_WAVE_CYCLES"] * 4 / (1024 * n["SQ_INSTS_VALU"] * clk), 3),
     "valu_wave_instructions": int(m["SQ_INSTS_VALU"]), "lds_wave_instructions": int(m["SQ_INSTS_LDS"]),
     "lds_bank_conflict_frac": round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 4),
     "lds_idx_active_frac_of_cu_cycles": round(m["SQ_LDS_IDX_ACTIVE"] / (256 * n["SQ_INSTS_VALU"] * clk), 4),
     "wait_any_per_wave_cycle": round(m.get("SQ_WAIT_ANY", 0) / max(m["SQ_WAVE_CYCLES"], 1), 4),
     "wait_inst_lds_per_wave_cycle": round(m.get("SQ_WAIT_INST_LDS", 0) / max(m["SQ_WAVE_CYCLES"], 1), 4),
     "hbm_read_bytes_x2": int(m.get("FETCH_SIZE", 0) * 2048), "hbm_write_bytes": int(m.get("WRITE_SIZE", 0) * 1024)}
print(json.dumps(d))
PY
