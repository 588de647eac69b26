OUT=gpurun_out/r05p; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_hm_dropin.py -m gpu -x -q -k "tencopencl or cpp or dropin or hm_ or reference_encoder or compat" 2>&1 | tail -4 | tee $OUT/tests.txt
for i in 1 2; do tools/class_latency; done | tee $OUT/class_latency.json
