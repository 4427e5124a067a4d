"""Dense step time for output widths around 8000 (the branch-free instances need a multiple of 4)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
n = 10000
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
for O in [int(a) for a in os.environ.get("OUT_WIDTHS", "8000 8001 8002 7999").split()]:
    p = f"/tmp/fdnn_out{O}.bin"
    if not os.path.exists(p):
        F.write_model_bin(p, F.synth_net([432] + [2048] * 7 + [O], seed=1))
    dnn = api.QuantizedDnn.loadFromFile(p)
    out = torch.empty((n, O), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(300): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dnn.profileBegin()
    for _ in range(50): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    print(O, {k: round(v["ms"] / 50, 4) for k, v in prof.items()})
    dnn.delete()
