#!/bin/bash
# Re-stamp roofline.traffic for the CURRENT library: the two PMC passes + gemm_hbm_traffic.json + the default bench line (a subset of
# collect_round_profiles.sh for when only the library changed).   usage (on the GPU box): bash tools/collect_traffic_only.sh r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-round}_traffic
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 0 --cpu-pairs 0 --no-roofline-events --no-vendor-comparator --no-clock-probe > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE/* $O/pmc_WRITE_SIZE/* > $O/pmc_traffic.txt 2>&1
python $R/tools/make_traffic_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 3 "profiles/${1:-round}_bench_pmc_hbm_traffic.txt" > $O/gemm_hbm_traffic.json 2>$O/gemm_hbm_traffic.err
cp $O/gemm_hbm_traffic.json $R/profiles/${1:-round}_gemm_hbm_traffic.json      # so that the bench line below reads the fresh stamp
python $R/bench.py > $O/default_run.json 2> $O/default_run.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -c 400 $O/default_run.json
