"""Per-kernel times for other topologies: NET="440 1024 1024 1024 1024 3483" FRAMES="500 10000" python tools/net_bench.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
topo = [int(a) for a in os.environ.get("NET", "440 1024 1024 1024 1024 3483").split()]
p = "/tmp/fdnn_net_" + "_".join(map(str, topo)) + ".bin"
if not os.path.exists(p):
    F.write_model_bin(p, F.synth_net(topo, seed=1))
dnn = api.QuantizedDnn.loadFromFile(p)
for n in [int(a) for a in os.environ.get("FRAMES", "500 10000").split()]:
    x = torch.from_numpy(F.synth_features(n, topo[0], seed=5)).cuda()
    out = torch.empty((n, topo[-1]), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(300): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dnn.profileBegin()
    for _ in range(50): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    ms = {k: round(v["ms"] / 50, 4) for k, v in prof.items()}
    macs = n * sum(a * b for a, b in zip(topo[1:-1], topo[2:]))
    print(topo, "n", n, ms, "int8 TOP/s over the int8 layers:", round(2 * macs / ((ms["hidden_gemm"] + ms["output_gemm"]) * 1e-3) / 1e12, 1))
