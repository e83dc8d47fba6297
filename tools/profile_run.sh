#!/bin/bash
# Evidence run (rounds 3 - 6: TG_PROFILE_TAG names the output set) on the MI355X box (one gpurun call): the default bench line (headline + other_configs + literal + CPU baseline),
# one bench line per env, rocprofv3 kernel stats and SQ counters of the default command and of object_push / object_balance /
# surface_follow-v2 (MG400, eight-sweep blocks) / the literal solver, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of
# edge_follow, object_push and object_balance.  Outputs under gpurun_out/<tag>/; copied to profiles/<tag>_* afterwards (tools/profile_collect.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TG_PROFILE_TAG:-r6_final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "import bench; print(bench.source_hash())" > $O/source_sha16.txt     # what every file of this run was measured on
timeout 500 python bench.py 2>$O/bench_edge.err | grep metric > $O/bench_edge.json
# the driver's own command (W = 5, K = 20), three times: what BENCH_rNN.json will hold; bench_driver_like.json = the first
for k in 1 2 3; do timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep metric > $O/bench_driver_like_$k.json; done
cp $O/bench_driver_like_1.json $O/bench_driver_like.json
B="timeout 200 python bench.py --no-cpu-baseline --no-companions"
$B --sync-steps --no-literal 2>/dev/null | grep metric > $O/bench_edge_syncsteps.json
$B --env surface_follow-v0 2>/dev/null | grep metric > $O/bench_surface_follow-v0.json
$B --env surface_follow-v2 2>/dev/null | grep metric > $O/bench_surface_follow-v2.json
$B --env object_balance-v0 --image-size 256 2>/dev/null | grep metric > $O/bench_object_balance-v0.json
$B --env object_push-v0 --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_object_push-v0.json
$B --env object_roll-v0 --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_object_roll-v0.json
$B --no-literal --observation-mode visuotactile --steps 200 --warmup 20 2>/dev/null | grep metric > $O/bench_edge_visuotactile.json
$B --no-literal --separate-policy 2>/dev/null | grep metric > $O/bench_edge_separate_policy.json
TG_RESET_BANK=0 $B --env surface_follow-v2 2>/dev/null | grep metric > $O/bench_surface_follow-v2_bank_off.json
$B --env object_push-v0 --narrowphase gjk_manifold --steps 100 --warmup 10 2>/dev/null | grep metric > $O/bench_object_push-v0_gjk_manifold.json
(timeout 100 python tools/pcie_rate.py; timeout 100 python tools/pcie_rate.py --tiles; timeout 100 python tools/pcie_rate.py --tiles; echo "TG_TILES_ZERO_COPY=0 (round 4: pack on the device, then copy):"; TG_TILES_ZERO_COPY=0 timeout 100 python tools/pcie_rate.py --tiles;
 echo "episodes out of phase (round 5):"; timeout 100 python tools/pcie_rate.py --tiles --staggered; timeout 100 python tools/pcie_rate.py --staggered) 2>&1 | grep -v amdgpu > $O/pcie_rate.txt
TG_FUSED_STEP=1 $B --no-literal 2>/dev/null | grep metric > $O/bench_edge_fused_step.json          # the one-launch step (opt-in: measured slower)
timeout 200 python tools/ball_rate.py 1024 8192 2>&1 | grep -v amdgpu > $O/ball_on_plate_rate.txt
timeout 200 python tools/spin_rate.py 1024 8192 2>&1 | grep -v amdgpu > $O/spin_rate.txt                      # object_balance spinning_plate (round 6)
# episodes out of phase (round 5): the rollout an RL run sees, reset bank auto (= on) and off, configs 2 and 3
(for e in edge_follow-v0 surface_follow-v0; do echo "$e, reset bank auto:"; timeout 100 python tools/desync_rate.py --env $e 2>&1 | grep aligned; echo "$e, TG_RESET_BANK=0:"; TG_RESET_BANK=0 timeout 100 python tools/desync_rate.py --env $e 2>&1 | grep aligned; done
 echo "object_push-v0 (1000-step episodes), reset template + optimistic move (round 6):"; timeout 200 python tools/desync_rate.py --env object_push-v0 --max-steps 1000 --steps 300 2>&1 | grep aligned
 echo "object_push-v0, TG_RESET_BANK=0 (optimistic move, no template):"; TG_RESET_BANK=0 timeout 200 python tools/desync_rate.py --env object_push-v0 --max-steps 1000 --steps 300 2>&1 | grep aligned
 echo "object_push-v0, TG_LITERAL_RESET=1 (round 5: every reset tick a contact tick):"; TG_LITERAL_RESET=1 timeout 200 python tools/desync_rate.py --env object_push-v0 --max-steps 1000 --steps 300 2>&1 | grep aligned) > $O/desync.txt
TG_BENCH_FORCE_COLLECTIVE=1 $B --no-literal --transport ipc --payload tiles 2>/dev/null | grep metric > $O/bench_edge_ipc_tiles_1rank.json
TG_NO_DIRECT_BATCH=1 TG_BENCH_FORCE_COLLECTIVE=1 $B --no-literal --transport ipc --payload tiles 2>/dev/null | grep metric > $O/bench_edge_ipc_tiles_1rank_copy_into_batch.json   # round 4's path: rank 0 copies its shard into the batch
TG_BENCH_FORCE_COLLECTIVE=1 $B --no-literal --transport collective --payload interior 2>/dev/null | grep metric > $O/bench_edge_rccl_interior_1rank.json
cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --no-cpu-baseline --no-literal --no-companions"
prof() {   # prof <name> <bench flags...>
    local name=$1; shift
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- $P "$@" > $O/prof_$name.log 2>&1
    find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$name.csv \;
    rm -rf $O/prof_$name
    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$name -- $P "$@" --steps 10 --warmup 2 > $O/pmc_$name.log 2>&1
    python $R/tools/pmc_parse.py $O/pmc_$name > $O/pmc_summary_$name.txt 2>&1
    rm -rf $O/pmc_$name
}
traffic() {   # traffic <name> <bench flags...>
    local name=$1; shift
    timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/tf_$name -- $P "$@" --steps 10 --warmup 2 > $O/tf_$name.log 2>&1
    timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/tw_$name -- $P "$@" --steps 10 --warmup 2 > $O/tw_$name.log 2>&1
    python $R/tools/traffic_parse.py $O/tf_$name $O/tw_$name > $O/traffic_$name.json 2>&1
    rm -rf $O/tf_$name $O/tw_$name
}
prof edge
prof edge_literal --full-sweeps --steps 200 --warmup 10
prof surface_follow-v2 --env surface_follow-v2 --steps 200 --warmup 10
prof object_push-v0 --env object_push-v0 --steps 100 --warmup 10
prof object_balance-v0 --env object_balance-v0 --image-size 256
prof surface_follow-v0 --env surface_follow-v0
traffic edge
traffic object_push-v0 --env object_push-v0
traffic object_balance-v0 --env object_balance-v0 --image-size 256
traffic surface_follow-v0 --env surface_follow-v0
rm -f $O/*.log
head -8 $O/kernel_stats_edge.csv
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
