"""Randomised parity run against the CPU oracle (test infrastructure): random topologies (input
width x4, hidden x16, any output width), weight scales from no saturating pairs to most pairs
saturating, both layer-0 flavours, random batch sizes across every kernel-selection branch, random masks.
Checked per case: every layer's u8 activations and int32 accumulators through the tap kernels
(bit-exact), the production kernels' last hidden layer (hiddenActivations, bit-exact), dense and
lazy soft-max (<= 2e-6), and the dense and masked results through the scoring loop (bit-identical to the calls).
  python tools/fuzz_parity.py [cases] [seed]      (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
tmp = os.environ.get("TMPDIR", "/tmp")
TIGHT = 2e-6
t0 = time.time()
worst = 0.0
for case in range(cases):
    in_dim = int(rng.choice([4, 8, 12, 40, 44, 64, 100, 132, 256, 432, 496]))  # (64 .. 496: the int8 screening of layer 0 from 640 frames up)
    hidden = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256, 272, 512, 1024]))
    n_hidden = int(rng.integers(3, 6))
    out = int(rng.choice([1, 3, 4, 31, 32, 33, 100, 257, 1000, 1001, 2048]))
    n = int(rng.choice([1, 2, 5, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 319, 320, 321, 400, 513, 700, 769,
                        1000, 1025, 1500, 2047, 2048, 2049, 2600, 3300]))
    if hidden >= 512 and n > 1100:
        n = int(rng.choice([700, 1000, 1100]))  # keep the CPU oracle's share of a case to a second or two
    w_std = float(rng.choice([0.02, 0.05, 0.05, 0.2, 1.0]))  # 1.0 with cutoff 3: weights clipped to +-127, most pairs can saturate
    topo = [in_dim] + [hidden] * n_hidden + [out]
    seed = int(rng.integers(1, 1 << 30))
    path = os.path.join(tmp, "fdnn_fuzz.bin")
    F.write_model_bin(path, F.synth_net(topo, seed=seed, w_std=w_std, bias_std=float(rng.choice([0.1, 1.0]))))
    x = F.synth_features(n, in_dim, seed=seed + 1, pad_from=None) * np.float32(rng.choice([0.3, 1.0, 3.0]))
    masks = (rng.random((n, out)) < rng.choice([0.05, 0.4, 0.9])).astype(np.int8)
    tag = f"case {case}: topo {topo} n {n} w_std {w_std} seed {seed}"
    fma = bool(rng.random() < 0.25)  # the reference as its own -march=native Makefile builds it: fused layer 0
    tag += f" fma {fma}"
    Oracle.set_l0_fma(fma)
    orc = Oracle(path)
    dnn = api.QuantizedDnn.loadFromFile(path)
    dnn.setInputLayerFma(fma)
    want, wt = orc.calculate(x, taps=True)
    taps = dnn.forwardTaps(x)
    for k in ("u8_acts", "acc_hid", "acc_out"):
        assert np.array_equal(taps[k], wt[k]), (tag, k)
    got = dnn.calculate(x, 10)
    # logits past 88.7 overflow exp in the reference too (no max-subtraction, dnn.cc:536-543): inf / inf = NaN for those
    # entries, 0 for the rest of the row -- the same entries must be NaN here
    assert np.array_equal(np.isnan(want), np.isnan(got)), (tag, "NaN pattern of an overflowing soft-max")
    fin = ~np.isnan(want)
    err = float(np.abs(got[fin] - want[fin]).max()) if fin.any() else 0.0
    # Nets with extreme weights (w_std 1.0: logits of +-20 and more over 1000+ outputs) show a few 1e-6 to a few 1e-5 where
    # sane nets (w_std <= 0.05, the SURVEY 8(d) distribution) show 1e-7 -- measured: the logits are bit-equal, the REFERENCE's rows then sum to 1.000004 (SoftMax::apply
    # adds its 2048 exp values sequentially in fp32, dnn.cc:536-540) and this library's to 1.0000000 (fixed-order tree).
    # The bar is 1e-3.
    tol = TIGHT if w_std <= 0.05 else 2e-4
    assert err <= tol, (tag, "dense", err)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    assert np.array_equal(ctx.hiddenActivations(), wt["u8_acts"][-1]), (tag, "production hidden layers")
    lazy = ctx.calculateForOutputNodesBatch(masks)
    lwant = orc.lazy(x, masks)
    assert np.array_equal(np.isnan(lwant), np.isnan(lazy)), (tag, "NaN pattern, lazy")
    lfin = ~np.isnan(lwant)
    lerr = float(np.abs(lazy[lfin] - lwant[lfin]).max()) if lfin.any() else 0.0
    assert lerr <= tol, (tag, "lazy", lerr)
    ctx.delete()
    srv = api.ScoringServer(dnn, max(n, 64), 2)
    t, o = srv.submit(x)
    tm, om = srv.submit(x, masks)  # the lazy contract through the loop, coalesced with the dense request
    srv.wait(t)
    srv.wait(tm)
    assert np.array_equal(o, got, equal_nan=True), (tag, "scoring loop")
    assert np.array_equal(om, lazy, equal_nan=True), (tag, "scoring loop, masked")
    srv.close()
    dnn.delete()
    orc.close()
    Oracle.set_l0_fma(False)
    worst = max(worst, err, lerr)
    if case % 10 == 9:
        print(f"{case + 1} cases, worst soft-max error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: {cases} cases in {time.time() - t0:.0f} s, worst soft-max error {worst:.2e}")
