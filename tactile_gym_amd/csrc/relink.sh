#!/bin/bash
# Development aid: links libtactile_gym_hip.so from the objects already in ../lib (after compiling single translation units by hand) and marks
# every object fresh for build.sh's TG_INCREMENTAL rule.  The build of record is build.sh.
set -euo pipefail
cd "$(dirname "$0")"
L=${TG_OUT:-../lib}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/tg_raster.o $L/tg_noise.o $L/tg_api.o $L/tg_contact_wave.o $L/tg_scene.o $L/tg_exchange.o $L/tg_fused.o $L/tg_broadphase.o $L/tg_api_state.o $L/tg_api_ops.o $L/tg_spin.o -o $L/libtactile_gym_hip.so
touch $L/*.o
ls -la $L/libtactile_gym_hip.so
