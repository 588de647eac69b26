#!/bin/bash
# round 4: refinement kernel, default library against build variants on the three contents:  bash tools/r04_frac_variants2.sh <tag> <variant> ...
TAG=${1:-r04v}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
one() {   # lib-variant size bit-depth content
  local L=""; [ $1 = default ] || L="HMME_LIB=$V/libhmme_$1.so"
  echo -n "$1 $2 $3-bit $4: "
  env $L python tools/refine_rate.py $2 $3 $4 2> $OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"
}
for rep in 1 2; do
  for v in default "$@"; do
    for c in coherent mixed noise; do one $v 3840x2160 8 $c; done
    one $v 1920x1080 8 coherent
    one $v 3840x2160 10 coherent; one $v 3840x2160 10 noise
  done
done | tee $OUT/frac_variants.txt
