#!/bin/bash
# ON THE GPU BOX: build and run the native serving-shape bench (tools/serve_bench.cpp).
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/fast-dnn_amd/lib
g++ -O2 -std=c++17 -o /tmp/serve_bench $R/tools/serve_bench.cpp -L$L -lfast-dnn -Wl,-rpath,$L -lpthread || exit 1
python - <<PY
import sys; sys.path.insert(0, "$R")
from fast_dnn_amd import formats as F
F.ensure_model_file("/tmp/fdnn_net_seed1_gauss.bin", F.NET_TOPOLOGY, seed=1, mode="gauss")
PY
M=/tmp/fdnn_net_seed1_gauss.bin
for T in 1 4 8 16 32 64; do
  /tmp/serve_bench $M $T 200 100 percall
  /tmp/serve_bench $M $T 200 100 server 6400 3 100
done
/tmp/serve_bench $M 16 200 100 batcher 6400 3 100
/tmp/serve_bench $M 8 20 1000 percall
/tmp/serve_bench $M 8 20 1000 server 10240 3 100
