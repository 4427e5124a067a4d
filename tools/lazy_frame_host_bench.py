"""The per-frame JNI protocol through the host API: calculateForOutputNodes(mask) per frame (host mask in, host row out,
one synchronisation per frame), at three mask densities."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
n = 100
x = F.synth_features(n, 432, seed=5)
ctx = dnn.getNewLazyContext(n)
ctx.calculateUntilOutput(x)
for density in (0.05, 0.4, 1.0):
    masks = np.ones((n, 8000), np.int8) if density >= 1 else F.generate_masks_fast(n, 8000, density, 0.03, seed=11)
    best = 1e9
    for rep in range(5):
        ctx.currentVectorIndex = 0
        t0 = time.perf_counter()
        for i in range(n): ctx.calculateForOutputNodes(masks[i])
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    print(f"{int(density * 100):3d} % active: {best:6.1f} us per frame", flush=True)
ctx.delete()
