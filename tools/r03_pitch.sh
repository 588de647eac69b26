#!/bin/bash
# 16-bit search at search ranges with their own LDS pitch: bash tools/r03_pitch.sh <tag>
TAG=${1:-r03e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for sr in 16 32 48 64 96 128; do
  timeout -k 10 300 python bench.py --bit-depth 10 --search-range $sr --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_10bit_sr$sr.json 2> $OUT/bench_10bit_sr$sr.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_10bit_sr$sr.json")); print("10-bit 2160p SR $sr:", d["value"], "GSAD/s", d["ms_per_step"], "ms")
PY
done | tee $OUT/pitch.txt
