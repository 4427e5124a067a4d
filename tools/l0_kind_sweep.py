"""Layer-0 time per kernel choice (0 = library's choice, 1 = chain kernel, 2 = 64 x 64-tile kernel, 4 = int8 screening)
against the batch size, to calibrate launch_l0's choice.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
NMAX = 10240
big = torch.from_numpy(F.synth_features(NMAX, 432, seed=5)).cuda()
out = torch.empty((NMAX, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
sizes = [int(a) for a in os.environ.get("FRAMES", "8 100 128 200 256 384 512 640 768 1000 1280 1500 2000 2560 3000 4000 5000 6000 7500 9000 10000").split()]
for n in sizes:
    row = {}
    for kind in (0, 1, 2, 4):
        dnn.setInputLayerKernel(kind)
        for _ in range(30): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        dnn.profileBegin()
        for _ in range(20): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = dnn.profileEnd()
        row[kind] = round(prof["l0"]["ms"] / 20 * 1e3, 1)
    dnn.setInputLayerKernel(0)
    print(f"n={n:6d}  l0 us: auto {row[0]:7.1f}  chain {row[1]:7.1f}  tile64 {row[2]:7.1f}  int8 screening {row[4]:7.1f}", flush=True)
