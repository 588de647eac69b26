#!/bin/bash
# Per-kernel times of short headline-only bench runs: `bash tools/trace_quick.sh TAG "ENV=.. ENV=.." size [size ...]` (through gpurun).
# One rocprofv3 --kernel-trace --stats run per size (no counters); the per-kernel summaries land in gpurun_out/TAG/<size><suffix>_kernel_stats.csv
set -e
TAG=$1; ENVS=$2; shift 2
export TMPDIR=/tmp
for kv in $ENVS; do export "$kv"; done
SUF=$(echo "$ENVS" | tr -c 'A-Za-z0-9=\n' '_')
mkdir -p gpurun_out/$TAG
for s in "$@"; do
  out=gpurun_out/$TAG/trace_${s}_$SUF
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --size $s > gpurun_out/$TAG/bench_${s}_$SUF.json 2> $out.err
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/$TAG/${s}_${SUF}_kernel_stats.csv
  rm -rf $out
done
