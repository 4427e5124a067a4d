#!/bin/bash
# Variant of libfast-dnn.so that differs in fdnn_gemm.hip only: tools/build_gemm_variant.sh NAME "-DFDNN_FUSE_SPLIT0=4"
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../fast-dnn_amd/csrc"
OUT=../lib; V=$OUT/variant_$NAME; mkdir -p $V
FLAGS="-DFDNN_ABLATION -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-result -Wno-unused-value -Wno-pass-failed $EXTRA"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c fdnn_gemm.hip -o $V/fdnn_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libfast-dnn-$NAME.so $V/fdnn_gemm.o $OUT/fdnn_chain.o $OUT/fdnn_l0.o $OUT/fdnn_kernels.o $OUT/fdnn_small.o $OUT/fdnn_l0s.o \
  $OUT/fdnn_runtime.o $OUT/fdnn_server.o $OUT/fdnn_group.o $OUT/fdnn_model.o $OUT/fdnn_jni.o -ldl -lpthread
echo "$OUT/libfast-dnn-$NAME.so"
