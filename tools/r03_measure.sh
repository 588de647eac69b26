#!/bin/bash
# round-3 measurement pass on the GPU box (through gpurun): bash tools/r03_measure.sh <tag> [what ...]
# what: tests bench sizes seq war   (default: all)
TAG=${1:-r03b}; shift
WHAT=${*:-tests bench sizes seq war}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  echo "== full GPU suite"
  timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "rc=$?" >> $OUT/gpu_tests.txt; tail -5 $OUT/gpu_tests.txt
fi
if has bench; then
  echo "== bench default"
  timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
fi
if has sizes; then
  echo "== pairs per launch"
  for cfg in "1080p 1" "1080p 2" "1080p 4" "1080p 8" "720p 1" "720p 4" "720p 8" "720p 16" "1200p 1" "1200p 4" "2160p 1" "2160p 2"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --size $1 --pairs $2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$1_pairs$2.json 2> $OUT/bench_$1_pairs$2.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_$1_pairs$2.json")); print("$1 pairs $2:", d["value"], "GSAD/s", d["ms_per_step"], "ms")
PY
  done
  for cfg in "1080p 1" "1080p 4" "720p 1" "720p 8"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --size $1 --pairs $2 --bit-depth 10 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$1_10bit_pairs$2.json 2> $OUT/bench_$1_10bit_pairs$2.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_$1_10bit_pairs$2.json")); print("$1 10-bit pairs $2:", d["value"], "GSAD/s", d["ms_per_step"], "ms")
PY
  done
fi
if has seq; then
  echo "== sequence: config 4 from a YUV file, resident vs streamed"
  python - <<PY
import sys; sys.path.insert(0, "hm-opencl_amd")
from hmme import synth
synth.Sequence(3840, 2160, 64, seed=777).write_yuv("/tmp/seq2160.yuv")
synth.Sequence(1920, 1080, 64, seed=777).write_yuv("/tmp/seq1080.yuv")
synth.Sequence(3840, 2160, 32, seed=777, bit_depth=10).write_yuv("/tmp/seq2160_10.yuv")
PY
  run() { name=$1; shift; timeout -k 10 600 python tools/me_sequence.py "$@" > $OUT/seq_$name.json 2> $OUT/seq_$name.err; python - <<PY
import json
d=json.load(open("$OUT/seq_$name.json")); print("$name:", d["pairs_per_s"], "pairs/s", d["gsad_per_s"], "GSAD/s", d["seconds"], "s", d["rank0"]["stages"])
PY
  }
  B="--frames 64 --gop randomaccess --size 2160p --yuv /tmp/seq2160.yuv --repeat 2"
  run 2160p_resident $B
  run 2160p_stream $B --stream
  run 2160p_stream_dl $B --stream --download
  run 2160p_resident_refine_dl $B --refine --download
  run 2160p_stream_refine_dl $B --stream --refine --download
  B="--frames 64 --gop randomaccess --size 1080p --yuv /tmp/seq1080.yuv --repeat 2"
  run 1080p_resident_k1 $B
  run 1080p_resident_k4 $B --pairs-per-launch 4
  run 1080p_stream_k4_dl $B --stream --pairs-per-launch 4 --download
  B="--frames 32 --gop randomaccess --size 2160p --bit-depth 10 --search-range 128 --yuv /tmp/seq2160_10.yuv --repeat 2"
  run 2160p_10bit_sr128_resident $B
  run 2160p_10bit_sr128_stream_dl $B --stream --download
  rm -f /tmp/seq2160.yuv /tmp/seq1080.yuv /tmp/seq2160_10.yuv
fi
if has war; then
  echo "== two-stream refill test on a library built WITHOUT the write-after-read wait (must fail)"
  bash tools/build_variant.sh nowar -DHMME_TEST_NO_WAR_WAIT > $OUT/nowar_build.txt 2>&1
  HMME_LIB=$PWD/hm-opencl_amd/csrc/build/variants/libhmme_nowar.so timeout -k 10 300 python -m pytest tests/test_gpu_sequence.py -q -m gpu -k refill > $OUT/nowar_test.txt 2>&1
  echo "rc=$? (expected: 1)" >> $OUT/nowar_test.txt; tail -4 $OUT/nowar_test.txt
fi
echo done
