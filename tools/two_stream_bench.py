"""Whole 10 000-frame passes on ONE stream back to back against the same passes alternating over K streams
(fdnn_calculate_device is asynchronous: kernels of different streams may share the chip workgroup by workgroup)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
n = int(os.environ.get("N", "10000"))
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
for K in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]
    outs = [torch.empty((n, 8000), dtype=torch.float32, device="cuda") for _ in range(K)]
    torch.cuda.synchronize()
    reps = 120
    for warm in (1, 0):
        t0 = time.perf_counter()
        for i in range(reps):
            k = i % K
            dnn.calculate_device(x.data_ptr(), n, outs[k].data_ptr(), streams[k].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{K} stream(s): {dt * 1e6:8.1f} us per pass  {n / dt / 1e6:7.3f} M frames/s   results identical: {same}", flush=True)
