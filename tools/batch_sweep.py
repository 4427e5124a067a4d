"""Whole-call time and per-kernel-class time of fdnn_calculate_device against the batch size
(device-resident in/out).  Looks for non-monotonic frames/s: a mis-set tile / kernel threshold.
FRAMES="100 1000 ..." python tools/batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
NMAX = 20480
big = torch.from_numpy(F.synth_features(NMAX, 432, seed=5)).cuda()
out = torch.empty((NMAX, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
default = "8 32 64 100 128 200 256 400 512 700 1000 1024 1500 2000 2047 2048 2560 3000 4000 5000 5120 6000 7000 8000 9000 10000 10240 12000 15360 15361 20000"
for n in [int(a) for a in os.environ.get("FRAMES", default).split()]:
    reps = 400 if n <= 2048 else 100
    for _ in range(reps): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dnn.profileBegin()
    for _ in range(20): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    us = {k: round(v["ms"] / 20 * 1e3, 1) for k, v in prof.items() if v["launches"]}
    print(f"n={n:6d}  {dt * 1e6:8.1f} us/call  {n / dt / 1e6:7.3f} M frames/s  {dt * 1e9 / n:7.1f} ns/frame   {us}", flush=True)
