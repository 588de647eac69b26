#!/bin/bash
# A/B of the refinement kernel variants on one box: bash tools/r03_frac_ab.sh <tag>
TAG=${1:-r03d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
for rep in 1 2; do
for v in default fw2 fw2old fw3old; do
  for c in coherent mixed noise; do
    if [ $v = default ]; then L=""; else L="HMME_LIB=$V/libhmme_$v.so"; fi
    echo -n "$v $c: "; env $L python tools/refine_rate.py 3840x2160 8 $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'])"
  done
done
done | tee $OUT/frac_ab.txt
