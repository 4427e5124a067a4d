"""Decoder-sized lazy blocks, device pointers in and out: calculateForOutputNodesBatchDevice per call (the hidden
layers are computed once, outside the timed loop): the masked small-batch output kernel + the soft-max scale.
(Round 3 timed a row-by-row kernel that skips the masked-out nodes against this path with this script: DESIGN.md
section 6 has the table; the kernel is gone.)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
O = dnn.outputDimension()
for n in (1, 2, 4, 8, 16, 32):
    x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutputDevice(x.data_ptr(), 0)
    line = f"n={n:3d}"
    for density in (0.05, 0.4, 1.0):
        m = np.ones((n, O), np.int8) if density >= 1 else F.generate_masks_fast(n, O, density, 0.03, seed=11)
        md = torch.from_numpy(m).cuda()
        od = torch.empty((n, O), dtype=torch.float32, device="cuda")
        for _ in range(50): ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(200): ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        line += f"   {int(density * 100):3d} % active {best:6.1f} us"
    print(line, flush=True)
    ctx.delete()
