# in-encoder evidence of the round-5 library: the reference's own encoder with tools/hm_patch, HMME_GPU_FRAC=1, HM's own searches run beside every engine call
OUT=gpurun_out/r05z_hm; mkdir -p $OUT
python tools/hm_ab.py --size 832x480 --frames 9 --gop RA --only GPU_FRAC --verify --log $OUT/log.txt > $OUT/hm_ab_832x480_RA_gpufrac_verify.json 2>$OUT/err1.txt; echo rc=$?
python tools/hm_ab.py --size 832x480 --frames 9 --gop B --only GPU_FRAC --verify --hm-args "--Profile=main10 --InternalBitDepth=10" --log $OUT/log.txt > $OUT/hm_ab_832x480_B_10bit_gpufrac_verify.json 2>$OUT/err2.txt; echo rc=$?
python tools/hm_ab.py --size 832x480 --frames 9 --gop B --only GPU_FRAC --verify --fade 0.04 --hm-args "--WeightedPredP=1 --WeightedPredB=1" --log $OUT/log.txt > $OUT/hm_ab_832x480_B_wp_fade_gpufrac_verify.json 2>$OUT/err3.txt; echo rc=$?
python tools/hm_ab.py --size 832x480 --frames 9 --gop P4 --only GPU_FRAC --verify --log $OUT/log.txt > $OUT/hm_ab_832x480_P4_gpufrac_verify.json 2>$OUT/err4.txt; echo rc=$?
for f in $OUT/*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['runs'][-1]; print('$f'.split('/')[-1], r['engine_calls'], 'calls', r['verified'], 'verified', r['verify_mismatches'], 'mismatches', r['failed'], 'failed', r['engine_ms_per_call'], 'ms/call')"; done
