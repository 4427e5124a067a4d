import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
big = torch.from_numpy(F.synth_features(256, 432, seed=5)).cuda()
out = torch.empty((256, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    dnn.calculate_device(big.data_ptr(), 100, out.data_ptr(), s)
    torch.cuda.synchronize()
