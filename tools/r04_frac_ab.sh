#!/bin/bash
# round 4: refinement kernel, workgroups that walk the jobs (grid = what the chip holds) against one workgroup per job, at 2 and 3 waves
# per SIMD:  bash tools/r04_frac_ab.sh <tag> <variant> [<variant> ...]   (variants under build/variants; "default" = libhmme.so)
TAG=${1:-r04b}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
one() {   # lib-variant grid size bit-depth content
  local L=""; [ $1 = default ] || L="HMME_LIB=$V/libhmme_$1.so"
  local G=""; [ $2 = auto ] && G="HMME_FRAC_GRID=-1" || G="HMME_FRAC_GRID=$2"   # auto = as many job-walking workgroups as the chip holds (the default when this was written; now -1)
  echo -n "$1 grid=$2 $3 $4-bit $5: "
  env $L $G HMME_TRACE=1 python tools/refine_rate.py $3 $4 $5 2> $OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'])"
  grep -h "workgroups per CU" $OUT/err.txt | sort -u | tr '\n' ' '; echo
}
for rep in 1 2; do
for v in "$@"; do
  for g in auto 0 256 512 1024; do
    one $v $g 3840x2160 8 coherent
    one $v $g 3840x2160 8 noise
  done
  one $v auto 3840x2160 8 mixed; one $v 0 3840x2160 8 mixed
  one $v auto 3840x2160 10 coherent; one $v 0 3840x2160 10 coherent
  one $v auto 3840x2160 10 noise; one $v 0 3840x2160 10 noise
  one $v auto 1920x1080 8 coherent; one $v 0 1920x1080 8 coherent
  one $v auto 2560x1440 8 coherent; one $v 0 2560x1440 8 coherent
done
done | tee $OUT/frac_ab.txt
