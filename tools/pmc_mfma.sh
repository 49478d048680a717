#!/bin/bash
# MFMA-pipe utilisation per kernel of the default bench step (rocprofv3 PMC, counters only -- no trace domains).
# usage (on the GPU box): bash tools/pmc_mfma.sh        -> gpurun_out/pmc_mfma/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mfma
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -B1 -A4 "mfma" > $O/avail.txt
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/raw -- \
    python $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-roofline-events --no-vendor-comparator --no-clock-probe --no-other-configs > $O/raw.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $O/ops -- \
    python $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-roofline-events --no-vendor-comparator --no-clock-probe --no-other-configs > $O/ops.log 2>&1
python $R/tools/pmc_mfma_summary.py $O | tee $O/summary.txt
