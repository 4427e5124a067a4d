"""A stream of device-resident calls of one batch size (the small / mid-size kernels under the profiler):
    python tools/small_call.py FRAMES [CALLS]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fast_dnn_amd import api, formats as F
n = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_gauss.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(calls):
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(calls):
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print(f"{n} frames: {(time.perf_counter() - t0) / calls * 1e6:.1f} us per call")
