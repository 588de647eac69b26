set -e
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo bench done
for wv in 2 3; do for c in coherent mixed noise; do echo -n "waves=$wv $c: "; HMME_FRAC_WAVES=$wv python tools/refine_rate.py 3840x2160 8 $c 2>>$OUT/err.txt; done; done | tee $OUT/refine_waves.txt
for tp in 0 2 3 4; do echo -n "1080p tail_parts=$tp: "; E=""; [ $tp = 0 ] || E="HMME_TAIL_PARTS=$tp"; env $E python bench.py --size 1080p --no-cpu-baseline --steps 40 2>>$OUT/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done | tee $OUT/cfg2_tail.txt
