"""Per-task phase clocks of the chained hidden-layer kernel (needs a -DFDNN_CHAIN_CLK=1 build: FDNN_LIB=...):
   FRAMES=10000 python tools/chain_clocks.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
topo = [432] + [2048] * 7 + [8000]
p = "/tmp/fdnn_net_clk.bin"
if not os.path.exists(p):
    F.write_model_bin(p, F.synth_net(topo, seed=1, mode=os.environ.get("MODE", "gauss")))
dnn = api.QuantizedDnn.loadFromFile(p)
n = int(os.environ.get("FRAMES", "10000"))
x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
s = torch.cuda.current_stream().cuda_stream
api.set_chain(1, 1)
ctx = dnn.getNewLazyContext(n)
for _ in range(20): ctx.calculateUntilOutputDevice(x.data_ptr(), s)
torch.cuda.synchronize()
cap = 4096
ctx.chainClocks(cap)
ctx.calculateUntilOutputDevice(x.data_ptr(), s)
torch.cuda.synchronize()
r = ctx.chainClocks(cap, fetch=True)
print("tasks recorded", len(r))
blk = r[:, 0] >> 32; xcd = r[:, 1] >> 32; lay = (r[:, 1] & 0xffffffff) >> 16; nt = r[:, 1] & 0xffff
q = (r[:, 0] & 0xffffffff) >> 24
tc = r[:, 3:10]
ph = np.diff(tc, axis=1)  # setup, wait, kloop, epilogue math, store+drain, next/arrive
names = ["setup+earlyW", "wait", "A+k-loop", "epi math", "stores+drain", "draw+arrive"]
print("all tasks: mean cycles per phase:", {k: int(v) for k, v in zip(names, ph.mean(axis=0))}, "total", int((tc[:, 6] - tc[:, 0]).mean()))
for l in range(int(lay.max()) + 1):
    m = lay == l
    print(f"layer {l}: n {m.sum()}", {k: int(v) for k, v in zip(names, ph[m].mean(axis=0))}, "total", int((tc[m, 6] - tc[m, 0]).mean()), "wait max", int(ph[m, 1].max()))
print("tasks run on the XCD of their queue:", float((q == xcd).mean()))
t0 = tc[:, 0].min()
per_blk = {}
for b in np.unique(blk):
    m = blk == b
    per_blk[b] = (int(tc[m, 0].min() - t0), int(tc[m, 6].max() - t0), int(m.sum()))
v = np.array(list(per_blk.values()))
print("workgroups", len(per_blk), "first start spread", int(v[:, 0].max()), "last end min/max", int(v[:, 1].min()), int(v[:, 1].max()), "tasks per wg min/max", int(v[:, 2].min()), int(v[:, 2].max()))
land = r[:, 2] - tc[:, 2]
print("stage 0 landed after the wait: mean cycles", int(land.mean()), "by layer", [int(land[lay == l].mean()) for l in range(int(lay.max()) + 1)])
