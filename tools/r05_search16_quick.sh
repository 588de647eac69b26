# 16-bit search kernel after a change: the tests that run it, then BASELINE config 5 and the 10-bit SR 64 / 1080p rates
OUT=gpurun_out/r05_search16; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16bit or 10bit or weighted or bipred or fuzz or shift_free or tiles or beyond or search_ctu or tiny" 2>&1 | tail -3 | tee $OUT/tests.txt
for a in "--bit-depth 10 --search-range 128" "--bit-depth 10 --search-range 64" "--bit-depth 10 --search-range 64 --size 1920x1080" "--bit-depth 12 --search-range 128"; do echo -n "$a: "; python bench.py --no-cpu-baseline --steps 8 --warmup 2 $a 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('verified', {}).get('slots'))"; done | tee $OUT/rates.txt
