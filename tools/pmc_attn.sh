#!/bin/bash
# SQ busy / VALU / MFMA / LDS / wait counters of the attention kernel (tools/attn_bench.py).  Run on the GPU box: bash tools/pmc_attn.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_attn; mkdir -p $O
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/tools/attn_bench.py > $O/log$i.txt 2>&1 || echo "set $i failed: $(tail -2 $O/log$i.txt)"
done
python - <<'PY'
import glob,csv,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_attn/p*/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(float);cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'attn_fwd' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']]+=float(r['Counter_Value']);cnt[r['Counter_Name']]+=1
    for k in sorted(agg): print(f"{k:32s} per launch {agg[k]/cnt[k]:.4g}")
PY
