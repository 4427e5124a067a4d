"""A few passes with a measurement build of the role-split fused output kernel (FDNN_LIB=fast-dnn_amd/lib_<variant>/libfast-dnn.so):
the clock-instrumented build prints per-phase cycles; every build gets its output-layer time from the profiling scopes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
mode = os.environ.get("MODE", "nosat")
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fdnn_net_seed1_{mode}.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode=mode)
dnn = api.QuantizedDnn.loadFromFile(p)
n = int(os.environ.get("N", "10000"))
reps = int(os.environ.get("REPS", "2"))
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
api.set_ppo(1)
for _ in range(reps):
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
if reps > 2:
    dnn.profileBegin()
    for _ in range(reps): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    print(f"n {n} {mode}", os.environ.get("FDNN_LIB", "shipped"), {k: round(v["ms"] / reps * 1e3, 1) for k, v in prof.items() if v["launches"]}, flush=True)
