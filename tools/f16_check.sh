set -x
mkdir -p gpurun_out/f16
timeout 900 python -m pytest tests/test_f16_operands_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/f16/t_f16.log
timeout 600 python -m pytest tests/test_headline_parity_gpu.py -x -q -s -k "p_large" 2>&1 | tail -12 > gpurun_out/f16/t_plarge_f16.log
SC_PRELN_F16=0 timeout 600 python -m pytest tests/test_headline_parity_gpu.py -x -q -s -k "p_large" 2>&1 | tail -12 > gpurun_out/f16/t_plarge_bf16.log
for i in 1 2 3; do
  for m in 1 0; do
    SC_PRELN_F16=$m timeout 300 python bench.py --model large --steps 10 --warmup 3 --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('PRELN_F16=$m pass $i', d['ms_per_step'], d.get('clock',{}).get('sclk_mhz_under_load'), d.get('loss'))
" >> gpurun_out/f16/bench_large_ab.log
  done
done
cat gpurun_out/f16/*.log
