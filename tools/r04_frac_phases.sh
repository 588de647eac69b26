#!/bin/bash
# round 4: where a refinement job's time goes, phase by phase (timeline build: every job leaves its start, end and the duration of its
# seven barrier-separated phases in the cost table): tools/build_variant.sh tl -DME_FRAC_T_TIMELINE; bash tools/r04_frac_phases.sh <tag>
TAG=${1:-r04p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/hm-opencl_amd/csrc/build/variants
for a in "3840x2160 8 coherent" "3840x2160 8 mixed" "3840x2160 8 noise" "1920x1080 8 coherent" "3840x2160 10 coherent"; do
  echo -n "$a: "
  HMME_LIB=$V/libhmme_tl.so HMME_TIMELINE=1 python tools/refine_rate.py $a 2> $OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); t=d['refine_ms'].pop('timeline')
print(d['refine_ms'], 'span', t['kernel_span_us'], 'job mean/min/max/p95', t['job_us_mean_min_max_p95'], 'busy', t['busy_share']); print('   phases (us, mean):', t['phase_us_mean'])"
done | tee $OUT/frac_phases.txt
