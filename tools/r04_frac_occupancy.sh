#!/bin/bash
# round 4: how many waves of me_frac_kernel are resident per SIMD?  SQ_WAVE_CYCLES (quad-cycles of resident waves) against GRBM_GUI_ACTIVE,
# for the product build (3 waves per SIMD allowed, spills to scratch) and variants.   bash tools/r04_frac_occupancy.sh <tag> <variant>...
TAG=${1:-r04j}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
for v in "$@"; do
  if [ $v = default ]; then unset HMME_LIB; else export HMME_LIB=$V/libhmme_$v.so; fi
  for content in coherent mixed noise; do
    for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
      name=$(echo $pass | cut -d' ' -f1)
      rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${v}_${content}_$name -- python3 tools/refine_rate.py 3840x2160 8 $content > $OUT/${v}_${content}_$name.json 2> $OUT/${v}_${content}_$name.err
    done
  done
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for v in sys.argv[2:]:
    for content in ("coherent", "mixed", "noise"):
        c = collections.defaultdict(list); ns = {}
        for name in ("SQ_WAVES", "GRBM_GUI_ACTIVE"):
            for p in glob.glob(f"{out}/{v}_{content}_{name}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(p)):
                    if "me_frac_kernel" in r["Kernel_Name"]:
                        c[r["Counter_Name"]].append(float(r["Counter_Value"]))
                        ns.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        m = {k: sum(x) / len(x) for k, x in c.items()}
        if "SQ_WAVE_CYCLES" not in m or "GRBM_GUI_ACTIVE" not in m:
            print(v, content, "counters missing", sorted(m)); continue
        kns = sum(ns["SQ_WAVES"]) / len(ns["SQ_WAVES"]); kg = sum(ns["GRBM_GUI_ACTIVE"]) / len(ns["GRBM_GUI_ACTIVE"])
        clk = m["GRBM_GUI_ACTIVE"] / 8 / kg
        print(json.dumps({"variant": v, "content": content, "kernel_us": round(kns / 1e3, 1), "clock_ghz": round(clk, 3), "waves": m["SQ_WAVES"],
                          "avg_waves_per_simd": round(m["SQ_WAVE_CYCLES"] * 4 / (1024 * kns * clk), 3),
                          "valu_insts_per_wave": round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"]),
                          "valu_busy_frac": round(m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * kns * clk), 3) if "SQ_ACTIVE_INST_VALU" in m else None,
                          "wait_any_share": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3), "wait_inst_any_share": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3)}))
PY
