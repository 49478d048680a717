#!/bin/bash
# Kernel-trace summaries of the two training steps (P-base and C-base tails) + the default training bench line -> gpurun_out/$1/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-train}
mkdir -p "$O"
trace() {
    local n=$1; shift
    timeout 900 rocprofv3 --kernel-trace -d "$O/$n.d" -o $n -- python $R/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe "$@" > "$O/$n.json" 2> "$O/$n.err"
    local db=$(find "$O/$n.d" -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > "$O/${n}_stats.txt"
    [ -d "$O/$n.d" ] && rm -rf "$O/$n.d"
}
trace train --train
trace casc_train --model cascaded --train
python $R/bench.py --train --cpu-pairs 0 > "$O/train_run.json" 2>/dev/null
head -12 "$O/train_stats.txt" | cut -c1-110
