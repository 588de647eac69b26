#!/bin/bash
# refinement kernel A/B on one box (through gpurun): bash tools/frac_ab.sh TAG content [variant ...]
# per variant (default = the shipped library; others: hm-opencl_amd/csrc/build/variants/libhmme_<variant>.so): 3 x tools/refine_rate.py (times, CRC of
# the tables), the counters of tools/frac_counters.sh, and -- for variants named tl_* (built with -DME_FRAC_T_TIMELINE) -- the job timeline
TAG=$1; C=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset HMME_LIB; else export HMME_LIB=$PWD/hm-opencl_amd/csrc/build/variants/libhmme_$v.so; fi
  case $v in
    tl_*) HMME_TIMELINE=1 WARM=20 python3 tools/refine_rate.py 3840x2160 8 $C > $OUT/${C}_${v}_timeline.json 2>> $OUT/err.txt ;;
    *) for i in 1 2 3; do WARM=20 python3 tools/refine_rate.py 3840x2160 8 $C >> $OUT/${C}_${v}_times.jsonl 2>> $OUT/err.txt; done
       bash tools/frac_counters.sh $C $OUT/ctr_${C}_$v > $OUT/${C}_${v}_counters.json 2>> $OUT/err.txt
       rm -rf $OUT/ctr_${C}_$v ;;
  esac
done
