"""Where does the role-split kernel differ from the in-phase tiles?  (FDNN_PP_ONLY=<layer> limits it to one hidden layer)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_dnn_amd import api, formats as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_gauss.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
x = F.synth_features(n, 432, seed=7)
api.set_chain(0)
def hid(pp):
    api.set_pp(pp, 1)
    ctx = dnn.getNewLazyContext(n); ctx.calculateUntilOutput(x); h = ctx.hiddenActivations().copy(); ctx.delete(); return h
a, b = hid(0), hid(1)
d = a != b
print("n", n, "differing bytes", int(d.sum()), "of", d.size)
if d.any():
    rows = np.flatnonzero(d.any(1)); cols = np.flatnonzero(d.any(0))
    print("rows", rows[:10], "...", rows[-5:], "count", rows.size)
    print("cols", cols[:10], "...", cols[-5:], "count", cols.size)
    # per (half tile of 160 rows, 64-column group) counts
    nh = (n + 159) // 160
    m = np.zeros((nh, 32), int)
    for h in range(nh):
        blk = d[h * 160:(h + 1) * 160]
        m[h] = blk.reshape(blk.shape[0], 32, 64).sum((0, 2))
    np.set_printoptions(linewidth=250)
    print(m[:12])
    r0 = rows[0]; c0 = np.flatnonzero(d[r0])[:8]
    print("row", r0, "cols", c0, "want", a[r0, c0], "got", b[r0, c0])
    # rows within the first bad half tile: per 32-row block
    h0 = r0 // 160
    blk = d[h0 * 160:(h0 + 1) * 160]
    print("per 32-row block of half", h0, blk.reshape(5, 32, -1).sum((1, 2)))
    print("per row-in-block (first bad block)", blk[:32].sum(1))
