#!/bin/bash
O=gpurun_out/r06r; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_headline_parity_gpu.py tests/test_varlen_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -n 2 $O/tests.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $R/$O/tr.d -o tr -- python $R/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs > $R/$O/tr.json 2> $R/$O/tr.err
db=$(find $R/$O/tr.d -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/$O/tr_stats.txt; rm -rf $R/$O/tr.d
cd $R; grep -E "posconv|conv0|layernorm768|weighted_sum" $O/tr_stats.txt | cut -c1-130
AB_TESTS=0 tools/r06_ab.sh r06r r06a
