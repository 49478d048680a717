#!/bin/bash
# A/B: HIP stream priority of the image tower's side stream (0 default / -1 high), three interleaved passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sideprio; mkdir -p $O; cd $R
COMMON="--steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator --no-clock-probe"
for i in 1 2 3; do
  for pr in 0 -1; do
    for cfg in "base" "large --model large" "casc --model cascaded"; do
      set -- $cfg; n=$1; shift
      SC_SIDE_PRIORITY=$pr timeout 300 python bench.py $COMMON "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n side_priority=$pr pass $i', d['ms_per_step'], d.get('loss'))
" >> $O/ab.log
    done
  done
done
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" >> $O/ab.log 2>&1
cat $O/ab.log
