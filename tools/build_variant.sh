#!/bin/bash
# A variant of libscvx_hip.so with extra -D flags on the conic-solve translation unit (A/B runs: tools/bench_variants.py).
#   tools/build_variant.sh NAME "-DSCVX_STREAM_U=8" [TU]  ->  build/libscvx_NAME.so      (needs a prior full build: build/obj/*.o)
# TU: the translation unit the flags apply to (default scvx_batch; e.g. scvx_threedof, scvx_discretize)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; TU=${3:-scvx_batch}
CS=$ROOT/successiveconvexification_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $ROOT/include -I $CS $FLAGS -c $CS/$TU.hip -o $ROOT/build/obj/${TU}_$NAME.o
OBJS=$(ls $ROOT/build/obj/*.hip.o | grep -v "$TU")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/libscvx_$NAME.so $ROOT/build/obj/${TU}_$NAME.o $OBJS
echo built build/libscvx_$NAME.so
