OUT=gpurun_out/r05d; mkdir -p $OUT; V=hm-opencl_amd/csrc/build/variants
for rep in 1 2; do for v in nofair fair fair2; do
 for a in "--size 720p" "--size 1200p" "--size 2160p --bit-depth 10 --search-range 128 --steps 6" "--size 1080p --bit-depth 10" "--size 2160p --search-range 128 --steps 6" "--size 1080p --refs 4"; do echo -n "$v $a: "; HMME_LIB=$V/libhmme_$v.so python bench.py $a --no-cpu-baseline 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
 echo -n "$v ctu_latency: "; HMME_LIB=$V/libhmme_$v.so python tools/ctu_latency.py 2>>$OUT/err.txt
done; done | tee $OUT/bench_fair2.txt
