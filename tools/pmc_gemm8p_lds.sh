#!/bin/bash
# LDS-port occupancy of gemm8p_pers_kernel (round 5): SQ_LDS_IDX_ACTIVE (all LDS-array cycles), bank conflicts, LDS instruction counts, busy cycles.
# usage (GPU box): bash tools/pmc_gemm8p_lds.sh  -> gpurun_out/pmc_gemm8p_lds/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm8p_lds; mkdir -p $O
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  SC_GEMM_KERNEL_MODE=16 timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/tools/gemm_traffic_probe.py fc2:128000,768,3072,3072,0 qkv:128000,2304,768,768,0 > $O/log$i.txt 2>&1 || echo "set $i failed: $(tail -2 $O/log$i.txt)"
done
python - <<'PY' | tee $O/summary.txt
import glob,csv,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_gemm8p_lds/p*/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'gemm8p' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']][int(r['Dispatch_Id'])].append(float(r['Counter_Value']))
    for k in sorted(agg):
        ds=sorted(agg[k]); n=len(ds)//2
        a=sum(sum(agg[k][d]) for d in ds[:n])/max(1,n); b=sum(sum(agg[k][d]) for d in ds[n:])/max(1,len(ds)-n)
        print(f"{k:28s} fc2 per launch {a:.4g}   qkv per launch {b:.4g}")
PY
