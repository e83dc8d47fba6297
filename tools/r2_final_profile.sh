#!/bin/bash
# Round-2 evidence run on the MI355X box (one gpurun call): bench lines of every env, rocprofv3 kernel stats and SQ counters of the
# default bench command and of object_push / object_balance / object_roll, HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of edge_follow
# and object_push.  Outputs under gpurun_out/r2_final/ (copied to profiles/r2_f_* afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TG_PROFILE_TAG:-r2_final}
mkdir -p $O
cd $R
python bench.py 2>/dev/null | grep metric > $O/bench_edge.json
python bench.py --no-cpu-baseline --sync-steps --no-literal 2>/dev/null | grep metric > $O/bench_edge_syncsteps.json
python bench.py --no-cpu-baseline --no-literal --num-envs 16384 2>/dev/null | grep metric > $O/bench_edge_16384.json
python bench.py --no-cpu-baseline --env surface_follow-v0 2>/dev/null | grep metric > $O/bench_surface_follow-v0.json
python bench.py --no-cpu-baseline --env object_balance-v0 --image-size 256 2>/dev/null | grep metric > $O/bench_object_balance-v0.json
python bench.py --no-cpu-baseline --env object_push-v0 --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_object_push-v0.json
python bench.py --no-cpu-baseline --env object_roll-v0 --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_object_roll-v0.json
python bench.py --no-cpu-baseline --no-literal --observation-mode visuotactile --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_edge_visuotactile.json
cd /tmp; export TMPDIR=/tmp
prof() {   # prof <name> <bench flags...>
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $R/bench.py --no-cpu-baseline --no-literal "$@" > $O/prof_$name.log 2>&1
    find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$name.csv \;
    rm -rf $O/prof_$name
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$name -- python $R/bench.py --no-cpu-baseline --no-literal "$@" --steps 10 --warmup 2 > $O/pmc_$name.log 2>&1
    python $R/tools/pmc_parse.py $O/pmc_$name > $O/pmc_summary_$name.txt 2>&1
    rm -rf $O/pmc_$name
}
traffic() {   # traffic <name> <bench flags...>
    local name=$1; shift
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/tf_$name -- python $R/bench.py --no-cpu-baseline --no-literal "$@" --steps 10 --warmup 2 > $O/tf_$name.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/tw_$name -- python $R/bench.py --no-cpu-baseline --no-literal "$@" --steps 10 --warmup 2 > $O/tw_$name.log 2>&1
    python $R/tools/traffic_parse.py $O/tf_$name $O/tw_$name > $O/traffic_$name.json 2>&1
    rm -rf $O/tf_$name $O/tw_$name
}
prof edge
prof object_push-v0 --env object_push-v0 --steps 100 --warmup 10
prof object_balance-v0 --env object_balance-v0 --image-size 256
prof object_roll-v0 --env object_roll-v0 --steps 100 --warmup 10
prof edge_visuotactile --observation-mode visuotactile --steps 100 --warmup 10
traffic edge
traffic object_push-v0 --env object_push-v0
traffic object_balance-v0 --env object_balance-v0 --image-size 256
head -8 $O/kernel_stats_edge.csv
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
