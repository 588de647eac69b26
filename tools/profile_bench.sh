#!/bin/bash
# Profiles `bench.py` with rocprofv3 on the GPU box.  Per configuration: one kernel-trace/--stats run and separate --pmc passes
# (never combined with other trace domains); every kernel of the engine that runs in the command is summarised -- the search
# kernel of the configuration and me_frac_kernel (bench.py's refinement leg).  Results under gpurun_out/prof_$TAG/<label>/,
# condensed into profiles/ by tools/summarize_profile.py.
# usage (through gpurun): bash tools/profile_bench.sh r02a            (all configurations)
#                         bash tools/profile_bench.sh r02a 8bit|10bit (one of them)
set -e
TAG=${1:-r02}
WHICH=${2:-all}
export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
        "GRBM_GUI_ACTIVE GRBM_COUNT")

profile_one() {   # label, bench arguments...
  local label=$1; shift
  local out=gpurun_out/prof_$TAG/$label
  mkdir -p $out
  echo "== $label: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/trace.err
  for pass in "${PASSES[@]}"; do
    name=$(echo $pass | cut -d' ' -f1)
    echo "== $label: pmc $name"
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_$name.err || echo "pass $name failed"
  done
  python3 tools/summarize_profile.py $out $TAG $label
}

# quick: the tooling itself on a small picture (tests/test_gpu_parity.py runs this: kernel trace + the SQ pass, summary written, latest_pmc_* untouched)
if [ "$WHICH" = quick ]; then
  PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT")
  profile_one quick_512x320_sr16 --size 512x320 --search-range 16
  exit 0
fi
if [ "$WHICH" = all ] || [ "$WHICH" = 8bit ]; then profile_one 2160p_sr64; fi
if [ "$WHICH" = all ] || [ "$WHICH" = 10bit ]; then profile_one 2160p_sr128_10bit --bit-depth 10 --search-range 128; fi
