"""Output widths off the 32 grid, fused against the scale pass: WIDTHS="8000 8001 8016 3483" python tools/outwidth_fused_bench.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
n = int(os.environ.get("FRAMES", "10000"))
for O in [int(a) for a in os.environ.get("WIDTHS", "8000 8001 8016 3483").split()]:
    p = f"/tmp/fdnn_w{O}.bin"
    if not os.path.exists(p):
        F.write_model_bin(p, F.synth_net([432] + [2048] * 7 + [O], seed=1))
    dnn = api.QuantizedDnn.loadFromFile(p)
    x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
    out = torch.empty((n, O), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in (0, 1):
        api.set_fuse(mode)
        for _ in range(100): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        dnn.profileBegin()
        for _ in range(50): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        pr = dnn.profileEnd()
        res[mode] = (pr["output_gemm"]["ms"] / 50 * 1e3, pr["normalize"]["ms"] / 50 * 1e3)
    api.set_fuse(-1)
    print(f"width {O}: scale pass {res[0][0]:.1f} + {res[0][1]:.1f} us, fused {res[1][0]:.1f} us", flush=True)
    dnn.delete()
