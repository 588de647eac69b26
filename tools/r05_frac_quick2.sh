OUT=gpurun_out/r05_frac_ride2; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -m gpu -x -q -k "frac or refine" 2>&1 | tail -2 | tee $OUT/tests.txt
for rep in 1 2; do for a in "3840x2160 8 coherent" "3840x2160 8 mixed" "3840x2160 8 noise" "1920x1080 8 coherent" "1920x1080 8 noise" "3840x2160 10 coherent" "3840x2160 10 noise"; do echo -n "$a: "; WARM=40 python tools/refine_rate.py $a 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'], d['tables_crc32'])"; done; done | tee $OUT/refine.txt
