"""Device-resident small calls back to back on one stream: us per call (the `small_batch` leg of bench.py uses the
same loop).  FRAMES="8 100" REPS=2000 python tools/small_call_bench.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
big = torch.from_numpy(F.synth_features(2048, 432, seed=5)).cuda()
out = torch.empty((2048, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "1000"))
for n in [int(a) for a in os.environ.get("FRAMES", "8 100").split()]:
    for _ in range(50): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:5d}  {dt * 1e6:8.1f} us/call", flush=True)
