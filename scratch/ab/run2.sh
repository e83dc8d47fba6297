#!/bin/bash
cd "$(dirname "$0")/../.."
for v in a b; do
  cp scratch/ab/lib_$v.so tactile_gym_amd/lib/libtactile_gym_hip.so
  for e in "--env edge_follow-v0" "--env surface_follow-v0" "--env object_balance-v0 --image-size 256"; do
    python bench.py --no-cpu-baseline --no-literal $e 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$e', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['k_step'], d['roofline']['kernel_ms']['k_render_tactile'])"
  done
done
cp scratch/ab/lib_b.so tactile_gym_amd/lib/libtactile_gym_hip.so
