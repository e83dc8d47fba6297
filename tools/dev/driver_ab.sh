#!/bin/bash
# development: the driver's command (W = 5, K = 20) with and without the untimed spin-up, three times each, then the default 2000-step line
for pw in 0 400 0 400 0 400 1500; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --pre-warm-ms $pw --no-companions --no-literal --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pre_warm', d['pre_warm_ms'], 'steps', d['pre_warm_steps'], 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'kernel_ms', r['kernel_ms'], 'win', r['profiled_window_ms_per_step'])"
done
python bench.py --gpus 1 --no-companions --no-literal --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('2000 steps: pre_warm', d['pre_warm_ms'], 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'kernel_ms', r['kernel_ms'], 'sum', r['kernel_ms_sum_per_step'], 'win', r['profiled_window_ms_per_step'])"
