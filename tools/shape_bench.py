import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
for n in (100, 1000, 10000):
    x = F.synth_features(n, 432, seed=5)
    masks = F.generate_masks(n, 8000, 0.40, 0.03, seed=11)
    ctx = dnn.getNewLazyContext(n)
    xd = torch.from_numpy(x).cuda(); md = torch.from_numpy(masks).cuda(); od = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    def step():
        ctx.calculateUntilOutputDevice(xd.data_ptr(), 0)
        ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
    try:
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"lazy-batch device n={n}: {dt*1e3:.3f} ms/step  {n/dt:,.0f} frames/s")
    except AttributeError as e:
        print("no device lazy api:", e); break
    # dense
    o2 = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    for _ in range(3): dnn.calculate_device(xd.data_ptr(), n, o2.data_ptr(), 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dnn.calculate_device(xd.data_ptr(), n, o2.data_ptr(), 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"dense device n={n}: {dt*1e3:.3f} ms/step  {n/dt:,.0f} frames/s")
    # per-frame lazy (JNI protocol): host round trip per frame
    ctx.calculateUntilOutput(x)
    t0 = time.perf_counter()
    for i in range(min(n, 100)): ctx.calculateForOutputNodes(masks[i])
    dt = (time.perf_counter() - t0) / min(n, 100)
    print(f"per-frame lazy call: {dt*1e6:.1f} us/frame")
    ctx.delete()
