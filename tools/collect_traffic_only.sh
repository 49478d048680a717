#!/bin/bash
# Re-stamp roofline.traffic for the CURRENT library: the two PMC passes + gemm_hbm_traffic.json + the default bench line (a subset of
# collect_round_profiles.sh for when only the library changed).   usage (on the GPU box): bash tools/collect_traffic_only.sh r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-round}_traffic
mkdir -p $O
# towers serialised in the PMC passes (kernels are serialised under counter collection anyway): the dispatch order is then the host's launch order, which
# make_traffic_json.py aligns with the launch list of the same command
export SC_OVERLAP_VIT=0
python $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs --dump-gemm-launches $O/gemm_launches.json > $O/launch_list_run.json 2> $O/launch_list_run.err
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 0 --cpu-pairs 0 --no-roofline-events --no-vendor-comparator --no-clock-probe --no-other-configs > $O/pmc_$c.log 2>&1
done
unset SC_OVERLAP_VIT
python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE/* $O/pmc_WRITE_SIZE/* > $O/pmc_traffic.txt 2>&1
python $R/tools/make_traffic_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 3 "profiles/${1:-round}_bench_pmc_hbm_traffic.txt" $O/gemm_launches.json > $O/gemm_hbm_traffic.json 2>$O/gemm_hbm_traffic.err
cp $O/gemm_hbm_traffic.json $R/profiles/${1:-round}_gemm_hbm_traffic.json      # so that the bench line below reads the fresh stamp
python $R/bench.py > $O/default_run.json 2> $O/default_run.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -c 400 $O/default_run.json
