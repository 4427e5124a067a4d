"""Layer-0 time and recomputed fraction per kernel kind (0 auto, 1 chain, 3 fp32 screening, 4 int8 screening) and batch size."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
NMAX = 10240
xh = F.synth_features(NMAX, 432, seed=5)
big = torch.from_numpy(xh).cuda()
out = torch.empty((NMAX, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
kinds = [int(k) for k in os.environ.get("KINDS", "4 3 1").split()]
for n in [int(a) for a in os.environ.get("FRAMES", "10000 4096 2048 1000").split()]:
    row = {}
    for kind in kinds:
        dnn.setInputLayerKernel(kind)
        for _ in range(30): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        dnn.profileBegin()
        for _ in range(30): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = dnn.profileEnd()
        _, rec = dnn.layer0(xh[:n])
        row[kind] = (round(prof["l0"]["ms"] / 30 * 1e3, 1), round(prof["fix"]["ms"] / 30 * 1e3, 1) if prof["fix"]["launches"] else 0.0, round(100.0 * rec / (n * 2048), 3))
    print(f"n={n:6d}  " + "  ".join(f"kind {k}: l0 {v[0]:6.1f} us fix {v[1]:5.1f} us recomputed {v[2]:.3f} %" for k, v in row.items()), flush=True)
