#!/bin/bash
# round 4: job-walking workgroups against one workgroup per job (HMME_FRAC_GRID=0) once BOTH deal the jobs last-first:
#   bash tools/r04_frac_grid_u16.sh <tag>
TAG=${1:-r04g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
one() {   # grid size bit-depth content
  local G=""; [ $1 = auto ] && G="HMME_FRAC_GRID=-1" || G="HMME_FRAC_GRID=$1"   # auto = as many job-walking workgroups as the chip holds (the default when this was written; now -1)
  echo -n "grid=$1 $2 $3-bit $4: "
  env $G python tools/refine_rate.py $2 $3 $4 2> $OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'])"
}
for rep in 1 2; do
  for g in auto 0; do
    for c in coherent mixed noise; do one $g 3840x2160 10 $c; done
    for c in coherent mixed noise; do one $g 3840x2160 8 $c; done
    one $g 1920x1080 10 coherent
  done
done | tee $OUT/frac_grid.txt
