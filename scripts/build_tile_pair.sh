#!/bin/bash
# Diagnostic library: conv_tile.hip built for TWO resident blocks per CU (<= 8 waves, 128 registers, 78 KB of LDS per block).
# Never shipped; scripts/tile_pair_probe.py measures what co-residency of two streams' blocks would buy.
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/micro
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEGAN_TILE_PAIR -Rpass-analysis=kernel-resource-usage -c ste_gan_amd/csrc/conv_tile.hip -o scripts/micro/conv_tile_pair.o 2>&1 | grep -E "VGPRs:|Scratch|Spill" | head -4
objs=$(ls ste_gan_amd/csrc/*.o | grep -v "\.f16\.o" | grep -v conv_tile.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libstegan_tile_pair.so $objs scripts/micro/conv_tile_pair.o
echo built scripts/micro/libstegan_tile_pair.so
