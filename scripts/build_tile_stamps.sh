#!/bin/bash
# Diagnostic library: conv_tile.hip with in-kernel time stamps (scripts/tile_stamps.py).  Never shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/micro
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEGAN_STAMPS -c ste_gan_amd/csrc/conv_tile.hip -o scripts/micro/conv_tile_stamps.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEGAN_STAMPS -c ste_gan_amd/csrc/conv_tile_wgrad.hip -o scripts/micro/conv_tile_wgrad_stamps.o
objs=$(ls ste_gan_amd/csrc/*.o | grep -v "\.f16\.o" | grep -v conv_tile.o | grep -v conv_tile_wgrad.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libstegan_tile_stamps.so $objs scripts/micro/conv_tile_stamps.o scripts/micro/conv_tile_wgrad_stamps.o
echo built scripts/micro/libstegan_tile_stamps.so
