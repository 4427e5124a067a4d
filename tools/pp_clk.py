"""One pass with the clock-instrumented role-split kernel (FDNN_LIB=fast-dnn_amd/lib_clk/libfast-dnn.so): per-phase cycles."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
mode = os.environ.get("MODE", "gauss")
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fdnn_net_seed1_{mode}.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode=mode)
dnn = api.QuantizedDnn.loadFromFile(p)
n = int(os.environ.get("N", "10000"))
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
api.set_chain(0); api.set_pp(1, 1)
for _ in range(2):
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
