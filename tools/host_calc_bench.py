import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
for n in (100, 1000, 10000):
    x = F.synth_features(n, 432, seed=5)
    for _ in range(3): dnn.calculate(x)
    t0 = time.perf_counter()
    for _ in range(10): dnn.calculate(x)
    dt = (time.perf_counter() - t0) / 10
    print(f"host-pointer calculate n={n}: {dt*1e3:.3f} ms  ({n/dt:,.0f} frames/s)")
