#!/bin/bash
# Evidence run on the MI355X box: bench lines + rocprofv3 kernel stats (+ optional PMC) for one env.
#   tools/gpu_profile.sh <tag> <env> [extra bench flags...]        outputs under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENV=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --env $ENV "$@" 2>$O/bench_$ENV.err | grep metric > $O/bench_$ENV.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$ENV -- python $R/bench.py --no-cpu-baseline --no-literal --env $ENV "$@" > $O/prof_$ENV.log 2>&1
find $O/prof_$ENV -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$ENV.csv \;
rm -rf $O/prof_$ENV
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$ENV -- python $R/bench.py --no-cpu-baseline --no-literal --env $ENV "$@" --steps 10 --warmup 2 > $O/pmc_$ENV.log 2>&1
python $R/tools/pmc_parse.py $O/pmc_$ENV > $O/pmc_summary_$ENV.txt 2>&1
rm -rf $O/pmc_$ENV
head -6 $O/kernel_stats_$ENV.csv; cat $O/pmc_summary_$ENV.txt; cat $O/bench_$ENV.json | head -c 600
