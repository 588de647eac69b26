OUT=gpurun_out/r05_frac_check; mkdir -p $OUT; V=$PWD/hm-opencl_amd/csrc/build/variants
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -m gpu -x -q -k "frac or refine" 2>&1 | tail -4 | tee $OUT/tests.txt
for rep in 1 2; do for c in coherent mixed noise; do echo -n "$c: "; WARM=40 python tools/refine_rate.py 3840x2160 8 $c 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"; done; done | tee $OUT/refine.txt
for a in "1920x1080 8 coherent" "3840x2160 10 coherent" "3840x2160 10 noise"; do echo -n "$a: "; WARM=40 python tools/refine_rate.py $a 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"; done | tee -a $OUT/refine.txt
for a in "3840x2160 8 coherent"; do echo -n "timeline $a: "; WARM=40 HMME_LIB=$V/libhmme_ftl.so HMME_TIMELINE=1 python tools/refine_rate.py $a 2>>$OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); t=d['refine_ms'].pop('timeline')
print(d['refine_ms'], 'span', t['kernel_span_us'], 'job mean/min/max/p95', t['job_us_mean_min_max_p95'], 'last_start', t['last_start_us'], t['latest_ends(job,start_us,dur_us,end_us)'][:4]); print('   phases (us, mean):', t['phase_us_mean'])"; done | tee $OUT/frac_phases.txt
