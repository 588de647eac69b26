mkdir -p gpurun_out/r06n
for rep in 1 2; do for s in 720p 1440p 1200p 1600p 832x480 2560x1088; do for m in 1 2; do echo -n "$s launches=$m " >> gpurun_out/r06n/ab.txt; HMME_TAIL_LAUNCHES=$m python bench.py --size $s --no-cpu-baseline --steps 20 2>>gpurun_out/r06n/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])" >> gpurun_out/r06n/ab.txt; done; done; done
cat gpurun_out/r06n/ab.txt
