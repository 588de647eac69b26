#!/bin/bash
# round 4: the 8-bit refinement kernel's two builds (HMME_FRAC_WAVES=2: 230 VGPRs, no scratch; =3: 168 VGPRs + spills) by launch size:
#   bash tools/r04_frac_waves.sh <tag>
TAG=${1:-r04w2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for wv in 2 3 auto; do
    for a in "832x480 8 coherent" "1280x720 8 coherent" "1920x1080 8 coherent" "1920x1080 8 mixed" "1920x1080 8 noise" "2560x1440 8 coherent" "3840x2160 8 coherent" "3840x2160 8 noise"; do
      E=""; [ $wv = auto ] || E="HMME_FRAC_WAVES=$wv"
      echo -n "waves=$wv $a: "
      env $E python tools/refine_rate.py $a 2> $OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"
    done
  done
done | tee $OUT/frac_waves.txt
