#!/bin/bash
# L2 (TCC) hit rate of the GEMM shapes under different tile orders.  usage (on the GPU box): bash tools/pmc_gemm.sh "<shapes>" "<band values>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPES=${1:-"qkv fc1"}
for band in ${2:-"0 4"}; do
  O=$R/gpurun_out/pmc_gemm/band$band
  SC_GEMM_BAND=$band timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O -- python $R/tools/gemm_bench.py $SHAPES > $O.log 2>&1
  python - "$O" "$band" $SHAPES <<'PY'
import glob,csv,collections,sys
per=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm256' in r['Kernel_Name']: per[r['Counter_Name']].append((int(r['Dispatch_Id']),float(r['Counter_Value'])))
shapes=sys.argv[3:]
for i,sname in enumerate(shapes):
    d={c:sum(x[1] for x in sorted(v)[7*i+2:7*i+7])/5 for c,v in per.items()}
    print(f"band={sys.argv[2]} {sname}: hit {d['TCC_HIT_sum']:.3g} miss {d['TCC_MISS_sum']:.3g} hit-rate {d['TCC_HIT_sum']/(d['TCC_HIT_sum']+d['TCC_MISS_sum']):.3f} tcp->tcc rd {d['TCP_TCC_READ_REQ_sum']:.3g} ea_rd {d.get('TCC_EA0_RDREQ_sum',0):.3g}")
PY
done
