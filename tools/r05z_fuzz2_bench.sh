OUT=gpurun_out/r05z; mkdir -p $OUT
HMME_FUZZ_CASES=6000 HMME_FUZZ_SEED=915000 HMME_FUZZ_CTU=4000 HMME_FUZZ_BIG=16 HMME_FUZZ_SLOTS=40 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k fuzz --durations=3 2>&1 | tail -8 | tee $OUT/fuzz_extended2.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py > $OUT/bench_default_after_profile.json 2> $OUT/bench_default2.err; echo "bench2 rc=$?"
g++ -O2 -o tools/class_latency tools/class_latency.cpp -Iinclude -Lhm-opencl_amd/host -lhmme_host -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/host -Wl,-rpath,$PWD/hm-opencl_amd/csrc && tools/class_latency | tee $OUT/class_latency.json
