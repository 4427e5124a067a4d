import os, sys, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
x = F.synth_features(2560, 432, seed=21)
got = dnn.calculate(x)
np.save("/tmp/fuse_got.npy", got)
print("sum rows", got.sum(1)[:4], got.sum(1).min(), got.sum(1).max())
print("row0[:12]", got[0, :12])
print("row0[60:72]", got[0, 60:72])
print("row0[250:262]", got[0, 250:262])
print("row300[:12]", got[300, :12])
