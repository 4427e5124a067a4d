"""Concurrency stress of the host entry points: T threads x (fdnn_calculate of a 100-frame utterance, then a LazyContext:
calculateUntilOutput + the batched lazy call), every result compared with the one of the same call made alone."""
import os, sys, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
T = int(os.environ.get("T", "16")); rounds = int(os.environ.get("ROUNDS", "60")); n = int(os.environ.get("N", "100"))
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
O = dnn.outputDimension()
xs = [F.synth_features(n, 432, seed=300 + t) for t in range(T)]
masks = [F.generate_masks(n, O, 0.40, 0.03, seed=500 + t) for t in range(T)]
refs, lrefs = [], []
for t in range(T):
    refs.append(dnn.calculate(xs[t]).copy())
    lc = dnn.getNewLazyContext(n)
    lc.calculateUntilOutput(xs[t])
    lrefs.append(lc.calculateForOutputNodesBatch(masks[t]).copy())
    lc.delete()
go = threading.Barrier(T); bad = []
def caller(t):
    go.wait()
    for r in range(rounds):
        got = dnn.calculate(xs[t])
        if not np.array_equal(got, refs[t]): bad.append(("dense", t, r, int((got != refs[t]).sum())))
        lc = dnn.getNewLazyContext(n)
        lc.calculateUntilOutput(xs[t])
        lg = lc.calculateForOutputNodesBatch(masks[t])
        if not np.array_equal(lg, lrefs[t]): bad.append(("lazy", t, r, int((lg != lrefs[t]).sum())))
        lc.delete()
th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
[h.start() for h in th]; [h.join() for h in th]
print("bad", len(bad), bad[:6], flush=True)
