#!/bin/bash
# builds a variant of the WHOLE library (host builder, GPU builder, kernels, C ABI) into build/ab/<name>.so with extra compiler
# flags -- for experiments that change the index layout or the minimizer order, where builder and kernels must agree
# usage: tools/ab_build_all.sh <name> [extra flags...]      (run a variant with MZ_LIB=build/ab/<name>.so python3 bench.py ...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/build/ab/obj_$NAME
mkdir -p $O
cd $ROOT/mazu_amd/csrc
CXX="-O3 -std=c++17 -fPIC -pthread -Wall -Wno-unused-result"
DEV="-Xclang -target-feature -Xclang -packed-fp32-ops"
for f in mz_host.cpp mz_capi.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX "$@" -c -o $O/${f%.*}.o $f & done
for f in mz_device.hip mz_u2pos.hip mz_build_gpu.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $DEV "$@" -Rpass-analysis=kernel-resource-usage -c -o $O/${f%.*}.o $f 2> $O/${f%.*}.resource.txt & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/build/ab/$NAME.so $O/mz_host.o $O/mz_device.o $O/mz_capi.o $O/mz_u2pos.o $O/mz_build_gpu.o
grep -A9 "k2u_wave_kernelILj7951" $O/mz_device.resource.txt | grep -E "VGPRs:|Occupancy|Scratch"
