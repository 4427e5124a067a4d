"""Layer 0 + hidden layers per pass, one launch per layer against the chained kernel, over batch sizes:
   FRAMES="4097 5000 ..." python tools/chain_sweep.py      (measurement builds: FDNN_CHAIN_TILE=256|320 forces the chain's tile)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
NMAX = 20480
big = torch.from_numpy(F.synth_features(NMAX, 432, seed=5)).cuda()
s = torch.cuda.current_stream().cuda_stream
default = "4097 4500 5000 5120 5500 6000 6400 7000 7680 8000 8500 9000 9500 10000 10240 10241 10500 11000 12000 12800 13000 14000 15000 15360 15361 16000 18000 20000 20480"
for n in [int(a) for a in os.environ.get("FRAMES", default).split()]:
    res = []
    for mode in (0, 1, -1):
        api.set_chain(mode, 1 if mode == 1 else 0)
        ctx = dnn.getNewLazyContext(n)
        for _ in range(30): ctx.calculateUntilOutputDevice(big.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60): ctx.calculateUntilOutputDevice(big.data_ptr(), s)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 60 * 1e6)
        ctx.delete()
    print(f"n={n:6d}  per-layer {res[0]:7.1f} us  chained {res[1]:7.1f} us  default rule {res[2]:7.1f} us   chain/per-layer {res[1] / res[0]:.3f}", flush=True)
