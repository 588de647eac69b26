OUT=gpurun_out/r05z; mkdir -p $OUT
HMME_FUZZ_CASES=10000 HMME_FUZZ_SEED=2026105 HMME_FUZZ_CTU=6000 HMME_FUZZ_BIG=24 HMME_FUZZ_SLOTS=40 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -x -q -m gpu -k fuzz --durations=4 2>&1 | tail -9 | tee $OUT/fuzz_extended3.txt
