#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in b w4; do
  cp scratch/ab/lib_$v.so tactile_gym_amd/lib/libtactile_gym_hip.so
  python bench.py --no-cpu-baseline --no-literal $ARGS 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['k_step'], d['roofline']['kernel_ms']['k_render_tactile'])"
done; done
cp scratch/ab/lib_a.so tactile_gym_amd/lib/libtactile_gym_hip.so
