#!/bin/bash
# The evidence of a round in gpurun calls of at most 20 minutes each:  bash tools/evidence.sh <tag> <part>   (through gpurun; results under gpurun_out/<tag>/)
#   measure   GPU suite, the default bench line (twice, around the profiles), tools/profile_bench.sh (rocprofv3 kernel stats + --pmc summaries of both
#             headline configurations -> profiles/ via tools/summarize_profile.py), the N-rank rehearsals on this one GPU, other picture sizes
#   validate  extended randomised parity (HMME_FUZZ_*), per-CTU call latencies (C ABI and TEncOpenCL class), the reference's own encoder with the
#             engine inside and HM's own searches beside every call (tools/hm_ab.py --verify)
# (replaces the per-round one-off scripts tools/r03_* .. r05z_*: what they measured is in profiles/ and DESIGN_HISTORY.md)
TAG=$1; PART=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$PART" = measure ]; then
  rm -rf gpurun_out/prof_$TAG
  python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/gpu_tests.txt
  python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
  bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1; echo "profile rc=$?"
  python bench.py > $OUT/bench_default_second_run.json 2> $OUT/bench_default2.err; echo "bench2 rc=$?"
  python bench.py --gpus 2 --share-gpu --backend gloo --steps 10 --warmup 2 > $OUT/bench_2rank_gloo_rehearsal_one_gpu.json 2> $OUT/bench_2rank.err; echo "2-rank rc=$?"
  python bench.py --gpus 4 --share-gpu --backend gloo --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_4rank_gloo_rehearsal_one_gpu.json 2> $OUT/bench_4rank.err; echo "4-rank rc=$?"
  for s in 720p 1080p 1200p 1440p 1600p; do python bench.py --size $s --no-cpu-baseline --steps 20 > $OUT/bench_$s.json 2>> $OUT/bench_sizes.err; done
  python bench.py --bit-depth 10 --search-range 64 --no-cpu-baseline --steps 10 > $OUT/bench_2160p_10bit_sr64.json 2>> $OUT/bench_sizes.err
elif [ "$PART" = validate ]; then
  HMME_FUZZ_CASES=${FUZZ_CASES:-3000} HMME_FUZZ_SEED=${FUZZ_SEED:-99000} HMME_FUZZ_CTU=${FUZZ_CTU:-2000} HMME_FUZZ_BIG=${FUZZ_BIG:-12} HMME_FUZZ_SLOTS=40 \
    python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k fuzz --durations=3 2>&1 | tail -8 | tee $OUT/fuzz_extended.txt
  g++ -O2 -o tools/class_latency tools/class_latency.cpp -Iinclude -Lhm-opencl_amd/host -lhmme_host -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/host -Wl,-rpath,$PWD/hm-opencl_amd/csrc && tools/class_latency | tee $OUT/class_latency.json
  g++ -O2 -o tools/ctu_latency_cpp tools/ctu_latency.cpp -Iinclude -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/csrc && tools/ctu_latency_cpp | tee $OUT/ctu_latency.json
  python tools/hm_ab.py --size 832x480 --frames 9 --gop RA --only GPU_FRAC --verify --log $OUT/hm_log.txt > $OUT/hm_ab_832x480_RA_gpufrac_verify.json 2>$OUT/hm_err1.txt; echo rc=$?
  python tools/hm_ab.py --size 832x480 --frames 9 --gop B --only GPU_FRAC --verify --hm-args "--Profile=main10 --InternalBitDepth=10" --log $OUT/hm_log.txt > $OUT/hm_ab_832x480_B_10bit_gpufrac_verify.json 2>$OUT/hm_err2.txt; echo rc=$?
  python tools/hm_ab.py --size 832x480 --frames 9 --gop B --only GPU_FRAC --verify --fade 0.04 --hm-args "--WeightedPredP=1 --WeightedPredB=1" --log $OUT/hm_log.txt > $OUT/hm_ab_832x480_B_wp_fade_gpufrac_verify.json 2>$OUT/hm_err3.txt; echo rc=$?
  python tools/hm_ab.py --size 832x480 --frames 9 --gop P4 --only GPU_FRAC --verify --log $OUT/hm_log.txt > $OUT/hm_ab_832x480_P4_gpufrac_verify.json 2>$OUT/hm_err4.txt; echo rc=$?
  for f in $OUT/hm_ab_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['runs'][-1]; print('$f'.split('/')[-1], r['engine_calls'], 'calls', r['verified'], 'verified', r['verify_mismatches'], 'mismatches', r['failed'], 'failed', r['engine_ms_per_call'], 'ms/call')"; done
else
  echo "usage: bash tools/evidence.sh <tag> measure|validate"; exit 2
fi
