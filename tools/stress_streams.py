"""Diagnosis of the many-streams test: 8 caller threads x fdnn_calculate_device at N frames on their own streams (+ a
second model), every result compared with the single-stream one; a mismatch is described (rows / columns / values)."""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
n = int(os.environ.get("N", "10000")); T = int(os.environ.get("T", "8")); rounds = int(os.environ.get("ROUNDS", "50"))
two = os.environ.get("TWO", "1") == "1"
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
p2 = "/tmp/second_fused.bin"
F.write_model_bin(p2, F.synth_net([432, 512, 512, 512, 2048], seed=77))
dnn2 = api.QuantizedDnn.loadFromFile(p2)
O, O2 = dnn.outputDimension(), dnn2.outputDimension()
x = torch.from_numpy(F.synth_features(n, 432, seed=41)).cuda()
ref = torch.empty((n, O), dtype=torch.float32, device="cuda"); ref2 = torch.empty((n, O2), dtype=torch.float32, device="cuda")
dnn.calculate_device(x.data_ptr(), n, ref.data_ptr(), 0); dnn2.calculate_device(x.data_ptr(), n, ref2.data_ptr(), 0)
torch.cuda.synchronize()
NT = T + (1 if two else 0)
outs = [torch.empty((n, O), dtype=torch.float32, device="cuda") for _ in range(T)]
out2 = torch.empty((n, O2), dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(NT)]
go = threading.Barrier(NT); lock = threading.Lock(); nbad = [0]
def caller(t):
    s = streams[t]; go.wait()
    for r in range(rounds):
        t0 = time.perf_counter()
        if t < T: dnn.calculate_device(x.data_ptr(), n, outs[t].data_ptr(), s.cuda_stream)
        else: dnn2.calculate_device(x.data_ptr(), n, out2.data_ptr(), s.cuda_stream)
        s.synchronize()
        dt = time.perf_counter() - t0
        got, want = (outs[t], ref) if t < T else (out2, ref2)
        with torch.cuda.stream(s):
            eq = torch.equal(got, want)
        if not eq or dt > 0.05:
            with lock:
                nbad[0] += 1
                d = (got != want)
                rows = d.any(1).nonzero().flatten().cpu().numpy(); cols = d.any(0).nonzero().flatten().cpu().numpy()
                print(f"thread {t} round {r} dt {dt*1e3:.1f} ms: {int(d.sum())} values differ; rows {rows[:6]}..{rows[-3:] if len(rows) else ''} ({len(rows)}), cols {cols[:6]}..{cols[-3:] if len(cols) else ''} ({len(cols)})", flush=True)
                if len(rows):
                    r0, c0 = int(rows[0]), int(cols[0])
                    print("   sample got/want", got[r0, c0].item(), want[r0, c0].item(), "row sums", got[r0].sum().item(), want[r0].sum().item(), flush=True)
th = [threading.Thread(target=caller, args=(t,)) for t in range(NT)]
[h.start() for h in th]; [h.join() for h in th]
print("bad", nbad[0], "giveups", dnn.fuseGiveups(), dnn2.fuseGiveups(), flush=True)
