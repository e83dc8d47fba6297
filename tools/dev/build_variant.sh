#!/bin/bash
# development: build_variant.sh <name> <extra hipcc flags...>  ->  tactile_gym_amd/lib_<name>/libtactile_gym_hip.so with tg_api.hip compiled with the extra
# flags and every other object taken from tactile_gym_amd/lib/ (A/B builds of the lane-mapped step / reset kernels; load with TG_HIP_LIBRARY)
set -euo pipefail
cd "$(dirname "$0")/../../tactile_gym_amd/csrc"
name=$1; shift
OUT=../lib_$name
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c tg_api.hip -o "$OUT/tg_api.o"
L=../lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$L/tg_raster.o" "$L/tg_noise.o" "$OUT/tg_api.o" "$L/tg_contact_wave.o" "$L/tg_scene.o" "$L/tg_exchange.o" "$L/tg_fused.o" "$L/tg_broadphase.o" "$L/tg_api_state.o" "$L/tg_api_ops.o" "$L/tg_spin.o" -o "$OUT/libtactile_gym_hip.so"
cp "$L/libtactile_gym_hip_test.so" "$OUT/"
echo "built $OUT/libtactile_gym_hip.so"
