"""Host-pointer fdnn_calculate (what the Java calculate() reaches): pageable input, pageable result.
Two flavours of the result array: fresh every call (first-touch page faults are part of the copy,
numpy's np.empty) and reused (already faulted in, like a JVM float[] that `new` has zeroed)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F

p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
fp = C.POINTER(C.c_float)
for n in (100, 1000, 10000):
    x = F.synth_features(n, 432, seed=5)
    for _ in range(3):
        dnn.calculate(x)
    t0 = time.perf_counter()
    for _ in range(10):
        dnn.calculate(x)
    dt = (time.perf_counter() - t0) / 10
    out = np.zeros((n, dnn.outputDimension()), dtype=np.float32)
    call = lambda: api.lib().fdnn_calculate(dnn.nativeDnnHandle, x.ctypes.data_as(fp), n, 432, 10, out.ctypes.data_as(fp))
    assert call() == 0
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    dr = (time.perf_counter() - t0) / 10
    print(f"host-pointer calculate n={n}: fresh result array {dt*1e3:.3f} ms ({n/dt:,.0f} frames/s), reused {dr*1e3:.3f} ms ({n/dr:,.0f} frames/s)")
