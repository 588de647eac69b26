#!/bin/bash
# round 4: where a refinement job's time goes.  Timing-only builds (results wrong, by design): t_noitems evaluates nothing (what is left is
# set-up, work lists, winners and the launches), t_noatomics computes every distortion but adds ONE value per item (what goes away is the
# LDS accumulation).  1080p fits ONE round of workgroups (510 jobs on 768 slots): its launch time is one job's latency.
#   tools/build_variant.sh t_noitems -DME_FRAC_T_NOITEMS; tools/build_variant.sh t_noatomics -DME_FRAC_T_NOATOMICS
#   bash tools/r04_frac_breakdown.sh <tag>
TAG=${1:-r04d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/r04_frac_variants.sh default t_noatomics t_noitems | tee $OUT/frac_breakdown.txt
