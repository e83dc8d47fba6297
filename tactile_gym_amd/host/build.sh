#!/bin/bash
# Builds libtg_host.so: the host half of the tile-sparse observation download (plain C; see tg_host_tiles.c).
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p ../lib
${CC:-gcc} -O3 -std=gnu99 -shared -fPIC -Wall -Wextra -pthread tg_host_tiles.c -o ../lib/libtg_host.so
echo "built ../lib/libtg_host.so"
