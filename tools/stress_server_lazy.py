"""Concurrency stress of the scoring loop with mixed traffic: T threads, each round one dense submission, one bit-mask lazy
submission (fdnn_server_submit_lazy_bits) and one byte-mask submission of ragged lengths, every result compared with the one of
the same call made alone; then the same through the model batcher (fdnn_calculate / fdnn_calculate_lazy_bits).
T=16 ROUNDS=40 python tools/stress_server_lazy.py"""
import os, sys, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
T = int(os.environ.get("T", "16")); rounds = int(os.environ.get("ROUNDS", "40"))
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
O = dnn.outputDimension()
lens = [100, 37, 250, 1, 100, 640, 64, 100, 2, 511, 100, 100, 1500, 100, 33, 100]
xs = [F.synth_features(lens[t % 16], 432, seed=300 + t) for t in range(T)]
masks = [F.generate_masks_fast(len(x), O, 0.40, 0.03, seed=500 + t) for t, x in enumerate(xs)]
bits = [F.pack_mask_bits(m) for m in masks]
refs = [dnn.calculate(x).copy() for x in xs]
lrefs = [dnn.calculateLazy(x, bits=b).copy() for x, b in zip(xs, bits)]
srv = api.ScoringServer(dnn, 2048, 3, 100)
go = threading.Barrier(T); bad = []
def caller(t):
    go.wait()
    out = np.empty((len(xs[t]), O), dtype=np.float32)
    for r in range(rounds):
        out.fill(-1.0)
        tk, _ = srv.submit(xs[t], out=out); srv.wait(tk)
        if not np.array_equal(out, refs[t]): bad.append(("dense", t, r, int((out != refs[t]).sum())))
        out.fill(-1.0)
        tk, _ = srv.submitLazy(xs[t], bits[t], out=out); srv.wait(tk)
        if not np.array_equal(out, lrefs[t]): bad.append(("lazy-bits", t, r, int((out != lrefs[t]).sum())))
        if (r + t) % 4 == 0:
            out.fill(-1.0)
            tk, _ = srv.submit(xs[t], masks[t], out=out); srv.wait(tk)
            if not np.array_equal(out, lrefs[t]): bad.append(("lazy-bytes", t, r, int((out != lrefs[t]).sum())))
th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
[h.start() for h in th]; [h.join() for h in th]
print("scoring loop: bad", len(bad), bad[:6], srv.stats(), flush=True)
srv.close()
dnn.enableBatcher(2048, 3, 100)
go = threading.Barrier(T); bad2 = []
def caller2(t):
    go.wait()
    for r in range(rounds):
        g = dnn.calculate(xs[t])
        if not np.array_equal(g, refs[t]): bad2.append(("dense", t, r))
        g = dnn.calculateLazy(xs[t], bits=bits[t])
        if not np.array_equal(g, lrefs[t]): bad2.append(("lazy", t, r))
th = [threading.Thread(target=caller2, args=(t,)) for t in range(T)]
[h.start() for h in th]; [h.join() for h in th]
print("model batcher: bad", len(bad2), bad2[:6], flush=True)
dnn.delete()
sys.exit(1 if bad or bad2 else 0)
