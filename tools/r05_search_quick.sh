# both search kernels after a generator change: the search tests, then the headline / 1080p / config 5 rates (bench.py checks 12 CTUs of each against the oracle)
OUT=gpurun_out/r05_search; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or fuzz or tiles or tiny or tail or 16bit or 10bit or weighted or bipred" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do python bench.py 2>>$OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); c=d['configs']
print('headline', d['value'], d['ms_per_step'], '| 1080p', c['config2_1080p_sr64']['gsad_per_s'], '| config5', c['config5_2160p_10bit_sr128']['gsad_per_s'], '| predictors', c['with_random_predictors']['gsad_per_s'], '| seq', c['config4_2160p_randomaccess_64_pictures_one_gpu']['pairs_per_s'])"; done | tee $OUT/rates.txt
