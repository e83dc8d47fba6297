#!/bin/bash
# A/B of two builds of the library: scratch/ab/lib_a.so vs lib_b.so, N alternating bench runs each
cd "$(dirname "$0")/../.."
ARGS=${ARGS:---no-cpu-baseline}
for rep in 1 2 3; do
  for v in a b; do
    cp scratch/ab/lib_$v.so tactile_gym_amd/lib/libtactile_gym_hip.so
    python bench.py $ARGS 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['k_step'], d['roofline']['kernel_ms']['k_render_tactile'])"
  done
done
