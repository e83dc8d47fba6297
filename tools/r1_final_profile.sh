#!/bin/bash
# Round-end evidence run on the MI355X box: bench lines for the four BASELINE envs + rocprofv3 kernel stats of the default bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py 2>/dev/null | grep metric > $O/bench_edge.json
python bench.py --no-cpu-baseline --sync-steps --no-literal 2>/dev/null | grep metric > $O/bench_edge_syncsteps.json
python bench.py --no-cpu-baseline --env surface_follow-v0 2>/dev/null | grep metric > $O/bench_surface_follow-v0.json
python bench.py --no-cpu-baseline --env object_balance-v0 --image-size 256 2>/dev/null | grep metric > $O/bench_object_balance-v0.json
python bench.py --no-cpu-baseline --env object_push-v0 --steps 30 --warmup 5 2>/dev/null | grep metric > $O/bench_object_push-v0.json
python bench.py --no-cpu-baseline --no-literal --num-envs 16384 2>/dev/null | grep metric > $O/bench_edge_16384.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-literal > $O/prof_bench.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/domain_stats.csv \;
rm -rf $O/prof
head -8 $O/kernel_stats.csv
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
