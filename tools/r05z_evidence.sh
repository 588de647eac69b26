# final evidence of round 5 (one gpurun call): GPU suite, default bench line, rocprofv3 summaries of both configurations, extended fuzz, latencies
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; rm -rf gpurun_out/prof_r05z   # (on the box this is empty anyway; what comes back is MERGED into the build container's gpurun_out/: delete gpurun_out/prof_r05z there before a new run, tools/summarize_profile.py reads every trace it finds)
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/gpu_tests.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
bash tools/profile_bench.sh r05z > $OUT/profile.log 2>&1; echo "profile rc=$?"
python bench.py > $OUT/bench_default_after_profile.json 2> $OUT/bench_default2.err; echo "bench2 rc=$?"
HMME_FUZZ_CASES=3000 HMME_FUZZ_SEED=77000 HMME_FUZZ_CTU=2000 HMME_FUZZ_BIG=10 HMME_FUZZ_SLOTS=40 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k fuzz --durations=3 2>&1 | tail -8 | tee $OUT/fuzz_extended.txt
g++ -O2 -o tools/class_latency tools/class_latency.cpp -Iinclude -Lhm-opencl_amd/host -lhmme_host -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/host -Wl,-rpath,$PWD/hm-opencl_amd/csrc && tools/class_latency | tee $OUT/class_latency.json
g++ -O2 -o tools/ctu_latency_cpp tools/ctu_latency.cpp -Iinclude -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/csrc && tools/ctu_latency_cpp | tee $OUT/ctu_latency.json
