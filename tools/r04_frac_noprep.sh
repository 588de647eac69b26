#!/bin/bash
# round 4: whole-picture refinement launches without the job-table kernel in front (every workgroup derives its job) against the table (HMME_FRAC_JOB_TABLE=1) and
# against the previous library (variant prev3):  bash tools/r04_frac_noprep.sh <tag>
TAG=${1:-r04np}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
for rep in 1 2; do
  for mode in default table prev3; do
    for a in "3840x2160 8 coherent" "3840x2160 8 mixed" "3840x2160 8 noise" "1920x1080 8 coherent" "832x480 8 coherent" "3840x2160 10 coherent" "3840x2160 10 noise"; do
      E=""; [ $mode = table ] && E="HMME_FRAC_JOB_TABLE=1"; [ $mode = prev3 ] && E="HMME_LIB=$V/libhmme_prev3.so"
      echo -n "$mode $a: "
      env $E python tools/refine_rate.py $a 2> $OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"
    done
  done
done | tee $OUT/frac_noprep.txt
