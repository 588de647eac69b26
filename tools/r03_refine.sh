#!/bin/bash
# refinement kernel: parity subset + rates (+ PMC of the coherent case with "pmc").  bash tools/r03_refine.sh <tag> [pmc]
TAG=${1:-r03c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_hm_dropin.py -q -m gpu -k "refine or refinement or frac or bipred or biprediction or tencopencl" > $OUT/refine_tests.txt 2>&1; echo "rc=$?" >> $OUT/refine_tests.txt; tail -4 $OUT/refine_tests.txt
for c in coherent mixed noise; do python tools/refine_rate.py 3840x2160 8 $c; done > $OUT/refine_rate.txt 2>&1
python tools/refine_rate.py 3840x2160 10 coherent >> $OUT/refine_rate.txt 2>&1
python tools/refine_rate.py 3840x2160 10 noise >> $OUT/refine_rate.txt 2>&1
cat $OUT/refine_rate.txt
if [ "$2" = pmc ]; then
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 tools/refine_rate.py 3840x2160 8 coherent > /dev/null 2> $OUT/pmc_$name.err || echo "pass $name failed"
  done
  python3 - <<PY
import csv, glob, collections, json
res = {}
for path in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "me_frac_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
if res:
    res["lds_bank_conflict_frac"] = res.get("SQ_LDS_BANK_CONFLICT", 0) / max(res.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    res["wait_any_over_active_any"] = res.get("SQ_WAIT_ANY", 0) / max(res.get("SQ_ACTIVE_INST_ANY", 1), 1)
json.dump(res, open("$OUT/frac_pmc_coherent.json", "w"), indent=1)
print(json.dumps(res))
PY
fi
