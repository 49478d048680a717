#!/bin/bash
# A/B: where the image tower enters the launch sequence of the CASCADED model (default: with the step; "head": beside the keyword head + CLIP text tower)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vithead; mkdir -p $O; cd $R
for i in 1 2 3; do
  for m in "" head layer10; do
    for cfg in "cascaded" "base"; do
      SC_VIT_START=$m timeout 300 python bench.py --model $cfg --steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg VIT_START=[$m] pass $i', d['ms_per_step'], d.get('clock',{}).get('sclk_mhz_under_load'), d.get('loss'))
" >> $O/ab.log
    done
  done
done
cat $O/ab.log
