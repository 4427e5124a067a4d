"""Per-class time of a device-resident pass with the role-split fused output kernel on and off (alternating, same process)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fast_dnn_amd import api, formats as F
mode = os.environ.get("MODE", "gauss")
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fdnn_net_seed1_{mode}.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode=mode)
dnn = api.QuantizedDnn.loadFromFile(p)
sizes = [int(v) for v in os.environ.get("N", "10000").split(",")]
for n in sizes:
    x = torch.from_numpy(F.synth_features(n, 432, seed=5)).cuda()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for label, ppo in (("in-phase fused", 0), ("role-split fused", 1), ("in-phase fused", 0), ("role-split fused", 1)):
        api.set_ppo(ppo)
        for _ in range(30): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        dnn.profileBegin()
        for _ in range(40): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = dnn.profileEnd()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        print(f"n {n:6d} {mode} {label:17s}", {k: round(v["ms"] / 40 * 1e3, 1) for k, v in prof.items() if v["launches"]}, f"pass {e0.elapsed_time(e1) / 40 * 1e3:.1f} us", "give-ups", dnn.fuseGiveups(), flush=True)
