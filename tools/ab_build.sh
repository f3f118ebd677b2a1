#!/bin/bash
# builds an A/B variant of libmazu_amd.so into build/ab/<name>.so with extra compiler flags for mz_device.hip
# usage: tools/ab_build.sh <name> [extra flags...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/ab/obj_$NAME
cd $ROOT/mazu_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wall -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -Rpass-analysis=kernel-resource-usage -c -o $ROOT/build/ab/obj_$NAME/mz_device.o mz_device.hip 2> $ROOT/build/ab/obj_$NAME/resource.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/build/ab/$NAME.so mz_host.o $ROOT/build/ab/obj_$NAME/mz_device.o mz_capi.o mz_u2pos.o mz_build_gpu.o
grep -A9 "k2u_wave_kernelILj7951" $ROOT/build/ab/obj_$NAME/resource.txt | grep -E "VGPRs:|Occupancy|LDS"
