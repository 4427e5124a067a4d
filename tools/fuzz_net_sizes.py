"""Random batch sizes on the full 432 -> 7x2048 -> 8000 net against the CPU oracle (test infrastructure).
Frames are independent, so the oracle only has to score a random sample of each batch: the GPU scores all
n frames (n random in 1 .. 20 000: every layer-0 kernel choice, every GEMM tile shape, chunked calls), the
oracle 48 of them; last hidden layer u8 bit-exact (production kernels), dense and lazy soft-max <= 2e-6.
  python tools/fuzz_net_sizes.py [cases] [seed]      (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_gauss.bin")
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
orc = Oracle(p)
O = dnn.outputDimension()
NMAX = 20000
big = F.synth_features(NMAX, 432, seed=4242)
bigd = torch.from_numpy(big).cuda()
out = torch.empty((NMAX, O), dtype=torch.float32, device="cuda")
lz = torch.empty((NMAX, O), dtype=torch.float32, device="cuda")
TIGHT = 2e-6
t0 = time.time()
worst = 0.0
edges = [1, 31, 33, 320, 321, 768, 769, 1024, 1025, 1200, 1201, 2047, 2048, 2049, 2560, 2561, 3400, 3600, 4096, 4097, 5120, 6144, 6145,
         8192, 8193, 10000, 10240, 10241, 15360, 15361, 20000]
for case in range(cases):
    n = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, NMAX + 1))
    off = int(rng.integers(0, NMAX - n + 1))
    idx = np.unique(np.concatenate([rng.integers(0, n, size=44), [0, n - 1, n // 2, max(0, n - 2)]]))
    x = big[off:off + n]
    masks = torch.from_numpy(F.generate_masks_fast(n, O, 0.4, 0.03, seed=int(rng.integers(1, 1 << 30)))).cuda()
    s = torch.cuda.current_stream().cuda_stream
    dnn.calculate_device(bigd[off:off + n].data_ptr(), n, out.data_ptr(), s)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutputDevice(bigd[off:off + n].data_ptr(), s)
    ctx.calculateForOutputNodesBatchDevice(masks.data_ptr(), lz.data_ptr(), 0, n, s)
    torch.cuda.synchronize()
    hid = ctx.hiddenActivations()
    ctx.delete()
    want, wt = orc.calculate(x[idx], taps=True)
    tag = f"case {case}: n {n} offset {off}"
    assert np.array_equal(hid[idx], wt["u8_acts"][-1]), (tag, "last hidden layer")
    sel = torch.from_numpy(idx).cuda()
    err = float(np.abs(out[:n][sel].cpu().numpy() - want).max())
    lerr = float(np.abs(lz[:n][sel].cpu().numpy() - orc.lazy(x[idx], masks[sel].cpu().numpy())).max())
    assert err <= TIGHT and lerr <= TIGHT, (tag, err, lerr)
    rs = out[:n].sum(1)
    assert float((rs - 1).abs().max()) < 1e-4, (tag, "row sums")
    worst = max(worst, err, lerr)
    if case % 10 == 9:
        print(f"{case + 1} cases, worst soft-max error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"net-size fuzz ok: {cases} cases in {time.time() - t0:.0f} s, worst soft-max error {worst:.2e}")
