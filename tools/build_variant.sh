#!/bin/bash
# Build an experiment variant of libfast-dnn.so next to the product library:
#   tools/build_variant.sh NAME "-DFDNN_GEMM_DEBUG=2 -DFDNN_L0_DEBUG=1"
# -> fast-dnn_amd/lib/libfast-dnn-NAME.so ; select it with FDNN_LIB=<path> (api.py).
# Kernel-ablation builds give wrong results on purpose; bench them with FDNN_BENCH_NOCHECK=1.
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../fast-dnn_amd/csrc"
OUT=../lib; V=$OUT/variant_$NAME; mkdir -p $V
FLAGS="-DFDNN_ABLATION -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -Wno-unused-value -Wno-pass-failed $EXTRA"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c fdnn_gemm.hip -o $V/fdnn_gemm.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -fno-slp-vectorize -c fdnn_l0.hip -o $V/fdnn_l0.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c fdnn_kernels.hip -o $V/fdnn_kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c fdnn_small.hip -o $V/fdnn_small.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c fdnn_l0s.hip -o $V/fdnn_l0s.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libfast-dnn-$NAME.so $V/fdnn_gemm.o $V/fdnn_l0.o $V/fdnn_kernels.o $V/fdnn_small.o $V/fdnn_l0s.o $OUT/fdnn_chain.o $OUT/fdnn_pp.o $OUT/fdnn_ppo.o \
  $OUT/fdnn_runtime.o $OUT/fdnn_server.o $OUT/fdnn_group.o $OUT/fdnn_model.o $OUT/fdnn_jni.o -ldl -lpthread
echo "$OUT/libfast-dnn-$NAME.so"
