#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps in the bf16 conv kernel (never shipped: scripts/micro/ is scratch).
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/micro
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 ${STAMP_DEFS:--DSTEGAN_STAMPS} $STAMP_FLAGS -c ste_gan_amd/csrc/conv_bf16.hip -o scripts/micro/conv_bf16_stamps.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 ${STAMP_DEFS:--DSTEGAN_STAMPS} $STAMP_FLAGS -c ste_gan_amd/csrc/conv_wgrad_bf16.hip -o scripts/micro/conv_wgrad_bf16_stamps.o
objs=$(ls ste_gan_amd/csrc/*.o | grep -v "\.f16\.o" | grep -v conv_bf16.o | grep -v conv_wgrad_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libstegan_stamps.so $objs scripts/micro/conv_bf16_stamps.o scripts/micro/conv_wgrad_bf16_stamps.o
echo built scripts/micro/libstegan_stamps.so
