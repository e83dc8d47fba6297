#!/bin/bash
# Builds libtactile_gym_hip.so (the product) and libtactile_gym_hip_test.so (device self-tests, loaded by tests/ only) for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   tg_raster.hip, tg_scene.hip and tg_noise.hip are compiled with -ffp-contract=off (bit-exact raster specification);
#   tg_api.hip (configuration, creation, step / reset launches; + tg_api_state.hip, tg_api_ops.hip: the rest of the C ABI, sharing tg_ctx.hpp), tg_contact_wave.hip (wave-per-env contact solver) and tg_exchange.hip (multi-GPU payloads, integer only) with the default contraction (FMA); tg_narrow_test.hip (GJK / EPA self-test) switches contraction off by pragma; tg_fused.hip (step + render in one launch) keeps FMA for the physics and
#   takes the raster from tg_raster_dev.hpp, whose pragma switches contraction off for everything after it.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${TG_OUT:-../lib}      # TG_OUT: another output directory (audit / A-B builds: TG_EXTRA_FLAGS=... TG_OUT=../lib_audit, loaded with TG_HIP_LIBRARY)
mkdir -p "$OUT"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value ${TG_EXTRA_FLAGS:-}"
# TG_INCREMENTAL=1 (development): keep an object whose translation unit and every header are older than it
stale() {   # stale <object> <source>: 0 if the object must be rebuilt
    [ "${TG_INCREMENTAL:-0}" = "1" ] || return 0
    [ -f "$1" ] || return 0
    for f in "$2" *.hpp *.h ../../include/*.h; do [ "$f" -nt "$1" ] && return 0; done
    return 1
}
cc() {      # cc <name> <extra flags...>
    local name=$1; shift
    if stale "$OUT/$name.o" "$name.hip"; then rm -f "$OUT/$name.o"; $HIPCC $COMMON "$@" -c "$name.hip" -o "$OUT/$name.o"; fi
}
cc tg_raster -ffp-contract=off & p1=$!
cc tg_noise -ffp-contract=off & p2=$!
cc tg_api & p3=$!
cc tg_contact_wave & p4=$!
cc tg_scene -ffp-contract=off & p5=$!
cc tg_exchange & p6=$!
cc tg_narrow_test & p7=$!
cc tg_fused & p8=$!
cc tg_selftest -ffp-contract=off & p9=$!
cc tg_broadphase -ffp-contract=off & p10=$!
cc tg_api_state & p11=$!
cc tg_api_ops & p12=$!
cc tg_spin & p13=$!
wait $p1; wait $p2; wait $p3; wait $p4; wait $p5; wait $p6; wait $p7; wait $p8; wait $p9; wait $p10; wait $p11; wait $p12; wait $p13    # each wait returns its job's status: a failed translation unit fails the build (set -e)
$HIPCC --offload-arch=gfx950 -shared -fPIC "$OUT/tg_raster.o" "$OUT/tg_noise.o" "$OUT/tg_api.o" "$OUT/tg_contact_wave.o" "$OUT/tg_scene.o" "$OUT/tg_exchange.o" "$OUT/tg_fused.o" "$OUT/tg_broadphase.o" "$OUT/tg_api_state.o" "$OUT/tg_api_ops.o" "$OUT/tg_spin.o" -o "$OUT/libtactile_gym_hip.so"
# test infrastructure (include/tactile_gym_hip_test.h): device self-tests of the raster's division / block test and of the wave-mapped GJK / EPA
$HIPCC --offload-arch=gfx950 -shared -fPIC "$OUT/tg_narrow_test.o" "$OUT/tg_selftest.o" -o "$OUT/libtactile_gym_hip_test.so"
echo "built $OUT/libtactile_gym_hip.so $OUT/libtactile_gym_hip_test.so"
