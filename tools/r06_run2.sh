#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
for L in "" tools/ab/libr06a.so; do echo "conv0 lib=${L:-product}" | tee -a $O/conv0.txt; for i in 1 2; do SPEECHCLIP_HIP_LIB=$L python tools/conv0_bench.py 2>/dev/null | tee -a $O/conv0.txt; done; done
AB_TESTS=0 tools/r06_ab.sh r06f r06a r05
python bench.py --steps 10 --warmup 3 --no-other-configs > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['vendor_comparator']['ms_per_step'], d['parity_check'])
PY
