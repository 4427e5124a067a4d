"""Debug: controlled risky-pair patterns in the second int8 layer."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle
hid = int(os.environ.get("HID", "128")); n = int(os.environ.get("N", "32"))
x = F.synth_features(n, 432, seed=9)

def run(tag, setw):
    net = F.synth_net([432, hid, hid, hid, 300], seed=17)
    # first int8 layer: push activations to the extremes (lots of 0 / 255)
    net.layers[1].weights[:] *= 8.0
    L = net.layers[2]
    L.weights[:] = 0.01 * np.sign(L.weights)
    setw(L.weights)
    p = "/tmp/dbg2.bin"
    F.write_model_bin(p, net)
    o = Oracle(p)
    want, wt = o.calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    dnn.delete()
    ueq = [(t["u8_acts"][j] == wt["u8_acts"][j]).all() for j in range(3)]
    a, b = t["acc_hid"][1], wt["acc_hid"][1]
    bad = np.argwhere(a != b)
    print(f"{tag}: risky {o.risky_pairs(2)} u8 equal {ueq} acc_hid0 ok {(t['acc_hid'][0] == wt['acc_hid'][0]).all()} acc_hid1 mismatches {len(bad)} of {a.size}")
    if len(bad):
        A = t["u8_acts"][1].astype(np.int64); W = o.layer_wq(2).astype(np.int64)
        for f, nd in bad[:6]:
            p2 = A[f][0::2] * W[nd][0::2] + A[f][1::2] * W[nd][1::2]
            c = np.clip(p2, -32768, 32767) - p2
            exact = int((A[f] * W[nd]).sum())
            print(f"   f {f} node {nd}: gpu {a[f, nd]} want {b[f, nd]} exact {exact} total_c {c.sum()}  gpu-exact {a[f, nd] - exact}  nz pairs {np.nonzero(c)[0][:10].tolist()} c {c[np.nonzero(c)[0]][:10].tolist()}")
        print("   bad frames", np.unique(bad[:, 0])[:40].tolist(), "bad nodes", np.unique(bad[:, 1])[:70].tolist())

def one_pair(w): w[0, 0:2] = 0.5
def node0(w): w[0, :] = 0.5
def node0_neg(w): w[0, :] = -0.5
def pair0_all(w): w[:, 0:2] = 0.5
def two_nodes(w): w[0, :] = 0.5; w[1, :] = 0.5
def nodes_0_4(w): w[0, :] = 0.5; w[4, :] = 0.5
def nodes_0_32(w): w[0, :] = 0.5; w[32, :] = 0.5
def nodes_0_64(w): w[0, :] = 0.5; w[64, :] = 0.5
def everything(w): w[:] = 0.5
for tag, fn in [("one_pair", one_pair), ("node0", node0), ("node0_neg", node0_neg), ("pair0_all", pair0_all), ("two_nodes", two_nodes),
                ("nodes_0_4", nodes_0_4), ("nodes_0_32", nodes_0_32), ("nodes_0_64", nodes_0_64), ("everything", everything)]:
    run(tag, fn)
