"""Layer-0 time at 10 000 frames (fdnn_debug_layer0 is host-to-host; this times calculate_device and reads the l0 class)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
mode = os.environ.get("MODE", "gauss")
p = f"/tmp/fdnn_net_seed1_{mode}.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode=mode)
dnn = api.QuantizedDnn.loadFromFile(p)
n = int(os.environ.get("N", "10000"))
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(60): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
for rep in range(3):
    dnn.profileBegin()
    for _ in range(40): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    prof = dnn.profileEnd()
    print({k: round(v["ms"] / 40 * 1e3, 1) for k, v in prof.items() if v["launches"]}, flush=True)
if os.environ.get("CLK"):
    c = dnn.deviceCounters(32)
    for name, base in (("block 0 wave 0", 4), ("mid block wave 0", 16)):
        t = c[base:base + 12]
        print(name, "cycles: prologue", t[1], "k-loop end", t[2], "epilogue math", t[3], "sync", t[4], "end", t[5],
              "| chunk 6: group 1", t[7] - t[6], "wait+barrier", t[8] - t[7], "dma issue", t[9] - t[8], "frag reads", t[10] - t[9], "group 2", t[11] - t[10])
