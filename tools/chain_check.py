"""Chained hidden layers against one launch per layer: last hidden layer's bytes equal, times.
   FRAMES="10000 9000 5000 12000" python tools/chain_check.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
topo = [int(a) for a in os.environ.get("NET", "432 2048 2048 2048 2048 2048 2048 2048 8000").split()]
mode = os.environ.get("MODE", "gauss")
p = "/tmp/fdnn_net_" + "_".join(map(str, topo)) + mode + ".bin"
if not os.path.exists(p):
    F.write_model_bin(p, F.synth_net(topo, seed=1, mode=mode))
dnn = api.QuantizedDnn.loadFromFile(p)
reps = int(os.environ.get("REPS", "30"))
for n in [int(a) for a in os.environ.get("FRAMES", "10000").split()]:
    x = F.synth_features(n, topo[0], seed=5)
    dx = torch.from_numpy(x).cuda()
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for chain in (0, 1):
        api.set_chain(chain, 1 if chain else 0)
        ctx = dnn.getNewLazyContext(n)
        for _ in range(3): ctx.calculateUntilOutputDevice(dx.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): ctx.calculateUntilOutputDevice(dx.data_ptr(), s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        dnn.profileBegin()
        for _ in range(reps): ctx.calculateUntilOutputDevice(dx.data_ptr(), s)
        torch.cuda.synchronize()
        prof = dnn.profileEnd()
        res[chain] = (ctx.hiddenActivations().copy(), dt, prof["hidden_gemm"]["ms"] / reps)
        ctx.delete()
    same = np.array_equal(res[0][0], res[1][0])
    print(f"n {n}: bytes equal {same} | l0+hidden per pass: per-layer {res[0][1]*1e6:.1f} us, chained {res[1][1]*1e6:.1f} us | hidden (events): {res[0][2]*1e3:.1f} -> {res[1][2]*1e3:.1f} us | faults {dnn.deviceCounters(8)[3]}", flush=True)
    if not same:
        d = np.argwhere(res[0][0] != res[1][0])
        print("  first diffs", d[:5], "count", len(d))
