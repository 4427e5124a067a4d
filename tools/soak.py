"""Soak: many batch sizes through every kernel-selection branch, checked for determinism (same
input -> same bits) and soft-max rows summing to one.  tools/soak.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
rng = np.random.default_rng(7)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
big = torch.from_numpy(F.synth_features(20000, 432, seed=3)).cuda()
s = torch.cuda.current_stream().cuda_stream
t0 = time.time()
for it in range(iters):
    n = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 100, 127, 128, 129, 255, 500, 1000, 1999, 2560, 3000, 3499, 3501, 4096, 5000, 9999, 10000, 10241, 16000, 20000]))
    fma = bool(rng.integers(0, 2))
    dnn.setInputLayerFma(fma)
    x = big[:n]
    a = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    b = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    dnn.calculate_device(x.data_ptr(), n, a.data_ptr(), s)
    dnn.calculate_device(x.data_ptr(), n, b.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(a, b), (it, n, fma)
    rs = a.sum(1)
    assert float((rs - 1).abs().max()) < 1e-3, (it, n, fma, float(rs.min()), float(rs.max()))
print(f"soak ok: {iters} iterations in {time.time() - t0:.1f} s")
