#!/bin/bash
# A/B of refinement kernel variants on one box: bash tools/r03_frac_ab.sh <tag> <variant> [<variant> ...]   (variants under build/variants; "default" = libhmme.so)
TAG=${1:-r03d}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
VARIANTS="$*"
for rep in 1 2; do
for v in $VARIANTS; do
  for cfg in "8 coherent" "8 mixed" "8 noise" "10 coherent" "10 noise"; do
    bd=${cfg% *}; content=${cfg#* }
    if [ $v = default ]; then L=""; else L="HMME_LIB=$V/libhmme_$v.so"; fi
    echo -n "$v $bd-bit $content: "; env $L python tools/refine_rate.py 3840x2160 $bd $content 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'])"
  done
done
done | tee $OUT/frac_ab.txt
