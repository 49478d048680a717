cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_gemm/p$i -- python $R/tools/gemm_bench.py qkv conv1 > $R/gpurun_out/pmc_gemm/log$i.txt 2>&1
  echo "set $i rc=$?"
done
python - <<'PY'
import glob,csv,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_gemm/p*/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(float);cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'gemm256' not in r['Kernel_Name']: continue
        # distinguish the two shapes by grid? use Grid_Size / kernel name + LDS
        k=(r['Kernel_Name'][:60].split('(')[0][-40:], r['Counter_Name'])
        agg[k]+=float(r['Counter_Value']);cnt[k]+=1
    for k in sorted(agg): print(k[0],k[1],cnt[k],'%.4g'%(agg[k]/cnt[k]))
PY
