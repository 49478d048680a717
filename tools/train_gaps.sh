#!/bin/bash
# idle-gap analysis of the training step and the forward step (rocprofv3 kernel trace -> tools/rocpd_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gaps; mkdir -p $O
for cfg in "train --train" "fwd" "casc --model cascaded"; do
  set -- $cfg; n=$1; shift
  timeout 600 rocprofv3 --kernel-trace -d $O/$n.d -o $n -- python $R/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs --no-roofline-events "$@" > $O/$n.json 2> $O/$n.err
  db=$(find $O/$n.d -name "*.db" | head -1)
  echo "== $n"; python $R/tools/rocpd_gaps.py $db conv0_stats 3
  python -c "import json,sys; d=[json.loads(l) for l in open('$O/$n.json') if l.startswith('{')][-1]; print('bench ms_per_step', d['ms_per_step'])"
  rm -rf $O/$n.d
done > $O/gaps.txt 2>&1
cat $O/gaps.txt
