"""Soak: many batch sizes through every kernel-selection branch -- dense, masked (lazy contract) and
through the scoring loop -- checked for determinism (same input -> same bits), soft-max rows
summing to one, masked-out nodes reading one value per row, and all-ones masks == dense.
tools/soak.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fast_dnn_amd import api, formats as F
p = "/tmp/fdnn_net_seed1_gauss.bin"
F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
dnn = api.QuantizedDnn.loadFromFile(p)
rng = np.random.default_rng(7)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
big = torch.from_numpy(F.synth_features(20000, 432, seed=3)).cuda()
s = torch.cuda.current_stream().cuda_stream
t0 = time.time()
for it in range(iters):
    n = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 100, 127, 128, 129, 255, 500, 1000, 1999, 2560, 3000, 3499, 3501, 4096, 5000, 9999, 10000, 10241, 16000, 20000]))
    fma = bool(rng.integers(0, 2))
    dnn.setInputLayerFma(fma)
    x = big[:n]
    a = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    b = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    dnn.calculate_device(x.data_ptr(), n, a.data_ptr(), s)
    dnn.calculate_device(x.data_ptr(), n, b.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(a, b), (it, n, fma)
    rs = a.sum(1)
    assert float((rs - 1).abs().max()) < 1e-3, (it, n, fma, float(rs.min()), float(rs.max()))
    # masked leg: the same frames through the lazy contract (device pointers), twice, + all-ones == dense
    if it % 3 == 0:
        m = (torch.rand((n, 8000), device="cuda") < 0.4).to(torch.int8)
        ctx = dnn.getNewLazyContext(n)
        la = torch.empty_like(a); lb = torch.empty_like(a)
        ctx.calculateUntilOutputDevice(x.data_ptr(), s)
        ctx.calculateForOutputNodesBatchDevice(m.data_ptr(), la.data_ptr(), 0, n, s)
        ctx.calculateForOutputNodesBatchDevice(m.data_ptr(), lb.data_ptr(), 0, n, s)
        torch.cuda.synchronize()
        assert torch.equal(la, lb), ("masked", it, n, fma)
        assert float((la.sum(1) - 1).abs().max()) < 1e-3, ("masked sum", it, n)
        off = m == 0
        lo = torch.where(off, la, torch.full_like(la, float("inf"))).min(1).values
        hi = torch.where(off, la, torch.full_like(la, float("-inf"))).max(1).values
        assert bool((lo == hi).all()), ("masked-out nodes differ within a row", it, n)
        m.fill_(1)
        ctx.calculateForOutputNodesBatchDevice(m.data_ptr(), la.data_ptr(), 0, n, s)
        torch.cuda.synchronize()
        assert torch.equal(la, a), ("all-ones mask != dense", it, n, fma)
        ctx.delete()
    # scoring-loop leg: two submissions in flight == the single-stream result
    if it % 5 == 0:
        srv = api.ScoringServer(dnn, n, 2)
        sa = torch.empty_like(a); sb = torch.empty_like(a)
        t1 = srv.submit_device(x.data_ptr(), n, sa.data_ptr()); t2 = srv.submit_device(x.data_ptr(), n, sb.data_ptr())
        srv.wait(t2); srv.wait(t1)
        assert torch.equal(sa, a) and torch.equal(sb, a), ("server", it, n, fma)
        srv.close()
print(f"soak ok: {iters} iterations in {time.time() - t0:.1f} s")
