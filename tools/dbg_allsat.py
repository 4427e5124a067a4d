"""Debug: where does the dense-fix-list net differ from the oracle?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle
hid = int(os.environ.get("HID", "256")); n = int(os.environ.get("N", "700"))
net = F.synth_net([432, hid, hid, hid, 300], seed=17)
rng = np.random.default_rng(5)
for L in net.layers[1:]:
    L.weights[:] = rng.choice(np.array([-0.5, 0.5, 0.45, -0.48], np.float32), size=L.weights.shape)
p = "/tmp/allsat_small.bin"
F.write_model_bin(p, net)
x = F.synth_features(n, 432, seed=9)
want, wt = Oracle(p).calculate(x, taps=True)
dnn = api.QuantizedDnn.loadFromFile(p)
t = dnn.forwardTaps(x)
print("l0 u8 equal", (t["u8_acts"][0] == wt["u8_acts"][0]).all())
for name in ("acc_hid", "acc_out"):
    a, b = t[name], wt[name]
    if a.ndim == 2: a, b = a[None], b[None]
    for j in range(a.shape[0]):
        bad = np.argwhere(a[j] != b[j])
        print(name, j, "mismatches", len(bad), "of", a[j].size)
        if len(bad):
            fr, nd = bad[:, 0], bad[:, 1]
            d = (a[j] - b[j])[fr, nd]
            print("  frames%32", np.unique(fr % 32)[:40], " nodes%64", np.unique(nd % 64)[:70])
            print("  frames", np.unique(fr)[:20], "nodes", np.unique(nd)[:20])
            print("  diff min/max", d.min(), d.max(), "first", bad[:5].tolist(), d[:5].tolist())
            break
