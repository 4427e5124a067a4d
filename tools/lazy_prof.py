"""Per-kernel times of the batched lazy call (masked output layer) next to the dense call."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
n = 10000
x = F.synth_features(n, 432, seed=5)
for density in (0.4, 1.0):
    masks = F.generate_masks(n, 8000, density, 0.03, seed=11) if density < 1 else np.ones((n, 8000), dtype=np.int8)
    ctx = dnn.getNewLazyContext(n)
    xd = torch.from_numpy(x).cuda(); md = torch.from_numpy(masks).cuda(); od = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    def step():
        ctx.calculateUntilOutputDevice(xd.data_ptr(), 0)
        ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
    for _ in range(100): step()
    torch.cuda.synchronize()
    dnn.profileBegin()
    for _ in range(50): step()
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    print("mask density", density, {k: round(v["ms"] / 50, 4) for k, v in prof.items()})
    ctx.delete()
