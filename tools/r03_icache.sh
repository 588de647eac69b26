#!/bin/bash
# instruction-cache counters of the search kernels: bash tools/r03_icache.sh <tag>
TAG=${1:-r03s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "8bit" "10bit --bit-depth 10 --search-range 128"; do
  name=${cfg%% *}; args=${cfg#* }; [ "$args" = "$name" ] && args=""
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQC_TC_INST_REQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    p=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${name}_$p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > /dev/null 2> $OUT/${name}_$p.err || echo "pass $p failed"
  done
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for path in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    cfg = path.split("/")[2].split("_")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        for name in ("me_search_kernel", "me_search16_kernel", "me_frac_kernel"):
            if name + "<" in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[cfg + ":" + k][c] = sum(v) / len(v)
for k, c in res.items():
    if "SQC_ICACHE_REQ" in c:
        c["icache_miss_frac"] = c.get("SQC_ICACHE_MISSES", 0) / max(c["SQC_ICACHE_REQ"], 1)
json.dump(res, open("$OUT/icache.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
