#!/bin/bash
# Builds libtactile_gym_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   tg_raster.hip and tg_noise.hip are compiled with -ffp-contract=off (bit-exact raster specification);
#   tg_api.hip (physics, control, C ABI) with the default contraction (FMA).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../lib
mkdir -p "$OUT"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
rm -f "$OUT/tg_raster.o" "$OUT/tg_noise.o" "$OUT/tg_api.o"
$HIPCC $COMMON -ffp-contract=off -c tg_raster.hip -o "$OUT/tg_raster.o" & p1=$!
$HIPCC $COMMON -ffp-contract=off -c tg_noise.hip -o "$OUT/tg_noise.o" & p2=$!
$HIPCC $COMMON -c tg_api.hip -o "$OUT/tg_api.o" & p3=$!
wait $p1; wait $p2; wait $p3    # each wait returns its job's status: a failed translation unit fails the build (set -e)
$HIPCC --offload-arch=gfx950 -shared -fPIC "$OUT/tg_raster.o" "$OUT/tg_noise.o" "$OUT/tg_api.o" -o "$OUT/libtactile_gym_hip.so"
echo "built $OUT/libtactile_gym_hip.so"
