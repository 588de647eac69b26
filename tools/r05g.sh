OUT=gpurun_out/r05g; mkdir -p $OUT; V=$PWD/hm-opencl_amd/csrc/build/variants
for v in ftl_noprio ftl; do for wv in 2 3; do for a in "3840x2160 8 coherent" "3840x2160 8 mixed" "3840x2160 8 noise"; do
  echo -n "$v waves=$wv $a: "
  HMME_FRAC_WAVES=$wv HMME_LIB=$V/libhmme_$v.so HMME_TIMELINE=1 python tools/refine_rate.py $a 2>> $OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); t=d['refine_ms'].pop('timeline')
print(d['refine_ms'], 'span', t['kernel_span_us'], 'job mean/min/max/p95', t['job_us_mean_min_max_p95'], 'busy', t['busy_share'], 'last_start', t['last_start_us'], 'first_start mean/max/p95', t['first_job_start_us_mean_max_p95']); print('   phases (us, mean):', t['phase_us_mean'])"
done; done; done | tee $OUT/frac_phases.txt
