"""Layer-0 (and every other kernel's) time against the batch size: how much does the partial last
round of 128 x 128 tiles cost?  FRAMES="8192 10000 ..." python tools/l0_scan.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
big = torch.from_numpy(F.synth_features(20480, 432, seed=5)).cuda()
out = torch.empty((20480, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in os.environ.get("FRAMES", "4096 8192 10000 10240 12288 16384 20480").split()]:
    for _ in range(200): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dnn.profileBegin()
    for _ in range(50): dnn.calculate_device(big.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    ms = {k: round(v["ms"] / 50, 4) for k, v in prof.items() if v["launches"]}
    tiles = ((n + 127) // 128) * 16
    print(n, "l0 tiles", tiles, "rounds@512", round(tiles / 512, 2), ms, "l0 ns/frame", round(ms["l0"] * 1e6 / n, 2), flush=True)
