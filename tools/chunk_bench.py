"""Whole-call time of fdnn_calculate_device for very large batches (run once per FDNN_CHUNK_FRAMES setting)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
NMAX = 125000
big = torch.from_numpy(F.synth_features(NMAX, 432, seed=5)).cuda()
out = torch.empty((NMAX, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
print("FDNN_CHUNK_FRAMES =", os.environ.get("FDNN_CHUNK_FRAMES", "(default)"))
for n in [int(a) for a in os.environ.get("FRAMES", "15361 20000 20480 30720 40960 125000").split()]:
    reps = 20
    for _ in range(5): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:7d}  {dt * 1e6:9.1f} us/call  {n / dt / 1e6:7.3f} M frames/s  {dt * 1e9 / n:7.1f} ns/frame", flush=True)
