#!/bin/bash
# round 4 A/B helper: refinement time of library variants (build/variants/libhmme_<name>.so; "default" = libhmme.so) on the three contents
#   bash tools/r04_frac_atomics.sh <variant> [<variant> ...]
export TMPDIR=/tmp; V=$PWD/hm-opencl_amd/csrc/build/variants
for rep in 1 2; do for v in "$@"; do for cfg in "1920x1080 coherent" "3840x2160 coherent" "3840x2160 mixed" "3840x2160 noise"; do
 L=""; [ $v = default ] || L="HMME_LIB=$V/libhmme_$v.so"
 echo -n "$v $cfg: "; env $L python tools/refine_rate.py ${cfg% *} 8 ${cfg#* } 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['refine_ms'])"
done; done; done
