#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -n 3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --cpu-pairs 0 --no-other-configs > $O/share_gpu_2ranks.json 2> $O/share_gpu_2ranks.err; tail -c 300 $O/share_gpu_2ranks.json
timeout 1500 python tools/soak_test.py > $O/soak.txt 2>&1; tail -n 8 $O/soak.txt
