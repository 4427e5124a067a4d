"""Per-workgroup start/end times of the hidden-layer GEMM launches (FDNN_GEMM_DEBUG=192 build: every
workgroup prints blockIdx and s_memrealtime at entry and exit).  Run ON THE GPU BOX:
  FDNN_LIB=.../libfast-dnn-dbgts.so python tools/wg_timeline.py"""
import os, subprocess, sys, re, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from fast_dnn_amd import api, formats as F
    p = "/tmp/fdnn_net_seed1_gauss.bin"
    F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 10000
    x = torch.from_numpy(F.synth_features(n, 432, seed=1000)).cuda()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    for _ in range(12):
        dnn.calculate_device(x.data_ptr(), n, out.data_ptr(), 0)
    torch.cuda.synchronize()
    sys.exit(0)
r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)
rows = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"^B (\d+) (\d+) (\d+)$", r.stdout, re.M)]
print("records", len(rows))
# group into launches: sort by start time, a launch = 256 consecutive blocks
rows.sort(key=lambda t: t[1])
launches = [rows[i:i + 256] for i in range(0, len(rows) - 255, 256)]
prev_end = None
for L in launches[-12:]:
    t0 = min(a for _, a, _ in L); t1 = max(b for _, _, b in L)
    if prev_end is not None:
        print(f"   first instruction {(t0 - prev_end) * 0.01:5.1f} us after the previous launch's last s_endpgm")
    prev_end = t1
    starts = sorted(a - t0 for _, a, _ in L); ends = sorted(t1 - b for _, _, b in L); dur = sorted(b - a for _, a, b in L)
    tick = 0.01  # us per s_memrealtime tick (100 MHz)
    byx = collections.defaultdict(list)
    for b, a, e in L: byx[b & 7].append((e - a) * tick)
    print(f"launch span {(t1 - t0) * tick:6.1f} us | start skew p50 {starts[128] * tick:4.1f} max {starts[-1] * tick:4.1f} | "
          f"duration p10 {dur[25] * tick:5.1f} p50 {dur[128] * tick:5.1f} p90 {dur[230] * tick:5.1f} max {dur[-1] * tick:5.1f} | "
          f"idle before end p50 {ends[128] * tick:4.1f} max {ends[-1] * tick:4.1f} | per-XCD mean "
          + " ".join(f"{sum(v) / len(v):5.1f}" for _, v in sorted(byx.items())))
