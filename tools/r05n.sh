OUT=gpurun_out/r05n; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_n_rank or bench_starts or bench_collective or launch_modes" 2>&1 | tail -5 | tee $OUT/tests.txt
tools/class_latency | tee $OUT/class_latency.json
python bench.py --size 1080p --no-cpu-baseline --steps 5 > $OUT/b.json 2>$OUT/b.err; tail -2 $OUT/b.err
