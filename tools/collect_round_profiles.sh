#!/bin/bash
# Round-end evidence in one call (on the GPU box): kernel-trace summaries of the forward / training steps (P-base and C-base), the two
# HBM-traffic PMC passes, the MFMA-utilisation PMC passes and the default bench line.  Everything lands in gpurun_out/$1/; copy what is to
# be judged into profiles/.   usage: bash tools/collect_round_profiles.sh r1e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-round}
mkdir -p $O
trace() {   # name, bench args...
    local n=$1; shift
    timeout 900 rocprofv3 --kernel-trace -d $O/$n.d -o $n -- python $R/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs "$@" > $O/$n.json 2> $O/$n.err
    local db=$(find $O/$n.d -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/${n}_stats.txt
    rm -rf $O/$n.d
}
trace fwd
SC_OVERLAP_VIT=0 trace fwd_serial_towers
trace train --train
trace casc_fwd --model cascaded
trace casc_train --model cascaded --train
trace large_fwd --model large
trace varlen_fwd --varlen
export SC_OVERLAP_VIT=0    # PMC passes with the towers serialised: dispatch order = launch order, aligned with the launch list (make_traffic_json.py)
python $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs --dump-gemm-launches $O/gemm_launches.json > $O/launch_list_run.json 2> $O/launch_list_run.err
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 0 --cpu-pairs 0 --no-roofline-events --no-vendor-comparator --no-clock-probe --no-other-configs > $O/pmc_$c.log 2>&1
done
unset SC_OVERLAP_VIT
python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE/* $O/pmc_WRITE_SIZE/* > $O/pmc_traffic.txt 2>&1
python $R/tools/make_traffic_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 3 "profiles/${1:-round}_bench_pmc_hbm_traffic.txt" $O/gemm_launches.json > $O/gemm_hbm_traffic.json 2>$O/gemm_hbm_traffic.err
cp $O/gemm_hbm_traffic.json $R/profiles/${1:-round}_gemm_hbm_traffic.json      # the bench lines below quote the fresh stamp
python $R/bench.py > $O/default_run.json 2> $O/default_run.err
python $R/bench.py --train --cpu-pairs 0 > $O/train_run.json 2> $O/train_run.err
python $R/bench.py --varlen --cpu-pairs 0 > $O/varlen_run.json 2> $O/varlen_run.err
python $R/bench.py --model large --cpu-pairs 0 > $O/large_run.json 2> $O/large_run.err
python $R/bench.py --model large --varlen --audio-len 240000 --cpu-pairs 0 > $O/large_varlen_run.json 2> $O/large_varlen_run.err
python $R/bench.py --model cascaded --cpu-pairs 0 > $O/casc_run.json 2> $O/casc_run.err
SC_OVERLAP_VIT=0 python $R/tools/step_gemm_breakdown.py > $O/step_gemm_breakdown.txt 2>&1
tail -c 600 $O/default_run.json
