"""Per-kernel times of the batched lazy call (masked output layer) next to the dense call."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
O = int(os.environ.get("OUT_WIDTH", "8000"))
p = f"/tmp/fdnn_out{O}.bin"
if not os.path.exists(p):
    F.write_model_bin(p, F.synth_net([432] + [2048] * 7 + [O], seed=1))
dnn = api.QuantizedDnn.loadFromFile(p)
n = 10000
x = F.synth_features(n, 432, seed=5)
for density in (0.4, 1.0):
    masks = F.generate_masks(n, O, density, 0.03, seed=11) if density < 1 else np.ones((n, O), dtype=np.int8)
    ctx = dnn.getNewLazyContext(n)
    xd = torch.from_numpy(x).cuda(); md = torch.from_numpy(masks).cuda(); od = torch.empty((n, O), dtype=torch.float32, device="cuda")
    def step():
        ctx.calculateUntilOutputDevice(xd.data_ptr(), 0)
        ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
    for _ in range(100): step()
    torch.cuda.synchronize()
    dnn.profileBegin()
    for _ in range(50): step()
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    print("mask density", density, "width", O, {k: round(v["ms"] / 50, 4) for k, v in prof.items()})
    ctx.delete()
