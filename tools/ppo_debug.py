"""Does the role-split fused output kernel (fdnn_ppo.hip) give the in-phase fused kernel's bits?  Device-resident, the result
buffer poisoned before every pass (a row the kernel does not write must show).
usage: ppo_debug.py [frames] [mode: gauss|nosat]"""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fast_dnn_amd import api, formats as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mode = sys.argv[2] if len(sys.argv) > 2 else "gauss"
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_%s.bin" % mode)
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode=mode)
dnn = api.QuantizedDnn.loadFromFile(p)
x = torch.from_numpy(F.synth_features(n, 432, seed=7)).cuda()
s = torch.cuda.current_stream().cuda_stream

def run(ppo):
    api.set_ppo(ppo)
    out = torch.full((n, 8000), float("nan"), dtype=torch.float32, device="cuda")
    g0 = dnn.fuseGiveups()
    dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    return out.cpu().numpy(), dnn.fuseGiveups() - g0

b, gb = run(1)
a, ga = run(0)
b2, gb2 = run(1)
ai, bi = a.view(np.uint32), b.view(np.uint32)
d = ai != bi
print("n", n, mode, "differing words", int(d.sum()), "of", d.size, "| second pass", int((b2.view(np.uint32) != ai).sum()), "| row sums", float(b.sum(1).min()), float(b.sum(1).max()),
      "| give-ups: role split", gb, gb2, "in phase", ga)
if d.any():
    rows = np.flatnonzero(d.any(1)); cols = np.flatnonzero(d.any(0))
    print("rows", rows[:10], "...", rows[-5:], "count", rows.size)
    print("cols", cols[:10], "...", cols[-5:], "count", cols.size)
    nh = (n + 159) // 160
    m = np.zeros((nh, 32), int)
    for h in range(nh):
        blk = d[h * 160:(h + 1) * 160]
        m[h] = [blk[:, c * 256:(c + 1) * 256].sum() for c in range(32)]
    np.set_printoptions(linewidth=250)
    print("per (half, node tile) differing words, halves with any:")
    for h in range(nh):
        if m[h].any(): print(h, m[h])
    r0 = rows[0]; c0 = np.flatnonzero(d[r0])[:8]
    print("row", r0, "cols", c0, "want", a[r0, c0], "got", b[r0, c0])
    print("nan", int(np.isnan(b).sum()), "zeros", int((b == 0).sum()), "of which expected", int((a == 0).sum()))
sys.exit(1 if d.any() else 0)
