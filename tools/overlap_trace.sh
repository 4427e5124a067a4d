#!/bin/bash
# ON THE GPU BOX: kernel timeline of the server loop at depth 2 (does the soft-max scale of batch i
# run under layer 0 of batch i+1?).  tools/overlap_trace.sh [depth]
R=$(cd "$(dirname "$0")/.." && pwd)
DEPTH=${1:-2}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ovl_prof
cat > /tmp/ovl.py <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
n = 10000
x = torch.from_numpy(F.synth_features(n, 432, seed=1000)).cuda()
outs = [torch.empty((n, 8000), dtype=torch.float32, device="cuda") for _ in range(4)]
srv = api.ScoringServer(dnn, n, $DEPTH)
for i in range(60): srv.submit_device(x.data_ptr(), n, outs[i % $DEPTH].data_ptr())
srv.drain(); srv.close(); dnn.delete()
PY
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl_prof -o k -- python /tmp/ovl.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ovl_prof/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    nm = r["Kernel_Name"]
    short = "inv" if "softmax_inv" in nm else "norm" if "normalize" in nm else "l0img" if "l0_image" in nm else "l0" if "l0_chain" in nm else "l0mfma" if "l0_mfma" in nm else "l0fix" if "l0_fix" in nm else "xnorm" if "l0_xnorm" in nm else "out" if "qgemm" in nm and "true" in nm.split("qgemm_kernel<")[1].split(">")[0].replace(" ","").split(",")[4] else "hid" if "qgemm" in nm else None
    if short: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
ev = ev[len(ev) // 2:]   # steady state
t0 = ev[0][0]
for s, e, k in ev[:40]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  {k}")
import collections
d = collections.defaultdict(list)
for s, e, k in ev: d[k].append((e - s) / 1e3)
print({k: round(sum(v) / len(v), 1) for k, v in d.items()})
PY
