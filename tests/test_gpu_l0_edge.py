"""The int8 screening of the input layer AT THE EDGE of its error bound (parity gap 1).

InputActivations + AddBias + QuantizedSigmoid (dnn.cc:219-286, dnn.h:29-47): the layer's byte is the table entry of
round(100 lin).  The screening kernel (fdnn_l0s.hip) computes lin~ from exact integer digit products and recomputes an
output exactly only where a rounding boundary of 100 lin lies within its bound Dd of 100 lin~.  Random data puts one output
in 140 near a boundary, so a bound too small by a constant factor could survive the fuzz runs for a long time.  Here

* the net's biases are chosen FROM THE ORACLE'S OWN chain sums so that for the first frame every node's 100 lin sits on a
  tie or 1 / 4 / 16 float ulps on either side of one, at a place of the sigmoid table where the byte changes, positive and
  negative half-integers alike (round() is half away from zero); the other frames are that frame with perturbations of
  a few ulps, so thousands of outputs crowd the same boundaries;
* every byte of both screening kernels (fp32 matrix pipe = kind 3, int8 = kind 4) is compared with the oracle;
* the bound's claim |100 lin_ref - t~| <= Dd is asserted on EVERY output of these batches and of a plain random one, and
  the observed maximum of the ratio -- the slack -- is printed."""
import os

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

H = 256


def f32(a):
    return np.asarray(a, dtype=np.float32)


def ulp_step(v: np.float32, k: int) -> np.float32:
    """v moved by k float32 ulps."""
    i = f32(v).view(np.int32)
    i = np.int32(i + k) if v >= 0 else np.int32(i - k)
    return i.view(np.float32)


def edge_net(tmp_models, seed=3):
    """(model path, frames): biases solved so that frame 0 of `frames` puts every node's 100 lin at / next to a tie."""
    net = F.synth_net([432, H, H, H, 100], seed=seed)
    net.layers[0].bias[:] = 0.0
    p0 = os.path.join(tmp_models, f"edge_nobias_{seed}.bin")
    F.write_model_bin(p0, net)
    x0 = F.synth_features(1, 432, seed=seed + 50)
    _, t = Oracle(p0).calculate(x0, taps=True)
    S = t["l0_lin"][0].copy()  # (l0 + l1) + (l2 + l3) with a zero bias: the reference's own chain sums (x + 0 = x)
    offsets = [0, 1, -1, 4, -4, 16, -16]
    hundred = np.float32(100.0)
    bias = np.zeros(H, dtype=np.float32)
    hit = 0
    for i in range(H):
        h = (i * 37) % 301 - 150  # half-integers h + 0.5 in -150 .. 150: the steep part of the table (adjacent entries differ)
        target_t = np.float32(h + 0.5)  # exactly representable
        want_t = ulp_step(target_t, offsets[i % len(offsets)])
        b0 = np.float32(np.float64(want_t) / 100.0 - np.float64(S[i]))
        best = None
        for k in range(-64, 65):  # walk the bias by ulps until fl(100 fl(S + b)) is the value wanted (AddBias: one fp32 add)
            b = ulp_step(b0, k)
            tt = np.float32(np.float32(S[i] + b) * hundred)
            d = abs(int(tt.view(np.int32)) - int(want_t.view(np.int32)))
            if best is None or d < best[0]:
                best = (d, b)
            if d == 0:
                break
        bias[i] = best[1]
        hit += best[0] == 0
    assert hit > H * 0.6, hit  # (the rest: 100 x does not reach every float; they still sit within a few ulps of the tie)
    net.layers[0].bias[:] = bias
    p = os.path.join(tmp_models, f"edge_{seed}.bin")
    F.write_model_bin(p, net)
    # frames: frame 0 as is, then copies with one / a few elements moved by 1 .. 3 ulps
    n = 1408
    rng = np.random.default_rng(seed)
    x = np.repeat(x0, n, axis=0)
    for f in range(1, n):
        for _ in range(int(rng.integers(1, 4))):
            j = int(rng.integers(0, 429))
            x[f, j] = ulp_step(x[f, j], int(rng.integers(-3, 4)))
    return p, x


@pytest.mark.parametrize("kind", [3, 4])
def test_every_output_next_to_a_rounding_boundary(tmp_models, kind):
    p, x = edge_net(tmp_models)
    if kind == 3:
        x = np.concatenate([x, x[:704]])  # the fp32 screen starts at 2048 frames
    want, wt = Oracle(p).calculate(x, taps=True)
    lin = wt["l0_lin"]
    t_ref = f32(lin) * np.float32(100.0)
    near = np.abs(t_ref - (np.floor(t_ref) + 0.5)) < 1e-3
    assert near[0].mean() > 0.95 and near.mean() > 0.5  # the construction works: the batch crowds the boundaries
    dnn = api.QuantizedDnn.loadFromFile(p)
    dnn.setInputLayerKernel(kind)
    got, recomputed = dnn.layer0(x)
    bad = np.argwhere(got != wt["u8_acts"][0])
    assert bad.size == 0, (len(bad), bad[:8].tolist())
    assert recomputed >= 0.5 * x.shape[0] * H  # nearly everything had to take the exact path
    # and the whole net behind it
    assert np.abs(dnn.calculate(x) - want).max() <= 2e-6
    dnn.delete()


def slack(dnn, orc, x):
    got, t, dd, _ = dnn.layer0Screen(x)
    _, wt = orc.calculate(x, taps=True)
    assert np.array_equal(got, wt["u8_acts"][0])
    assert np.isfinite(t).all() and np.isfinite(dd).all() and (dd > 0).all()
    t_ref = (f32(wt["l0_lin"]) * np.float32(100.0)).astype(np.float64)
    ratio = np.abs(t_ref - t.astype(np.float64)) / dd.astype(np.float64)
    return float(ratio.max()), float(np.median(ratio)), float((ratio > 0.25).mean())


def test_the_bound_holds_on_every_output_and_its_slack(tmp_models, net_model_path):
    """|100 lin_ref - t~| / Dd over all outputs: <= 1 is the bound's claim (a value above 1 = an output the screen could
    pass wrongly).  Reported for the edge batch, a plain random batch on the full 432 -> 2048 layer, and wide-range rows."""
    p, x = edge_net(tmp_models, seed=7)
    dnn = api.QuantizedDnn.loadFromFile(p)
    r_edge = slack(dnn, Oracle(p), x)
    dnn.delete()
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    orc = Oracle(net_model_path)
    xr = F.synth_features(640, 432, seed=99)
    r_rand = slack(dnn, orc, xr)
    xs = xr.copy()
    xs[:, ::7] *= np.float32(40.0)   # a few dominant columns: the other elements lose digits of their 24-bit image
    xs[::3] *= np.float32(1e-3)
    r_wide = slack(dnn, orc, xs)
    dnn.delete()
    print(f"\nint8 screening, max / median of |100 lin_ref - t~| / Dd and share above 0.25: edge batch {r_edge}, random batch {r_rand}, wide-range rows {r_wide}")
    for r in (r_edge, r_rand, r_wide):
        assert r[0] <= 1.0, r


def test_flagged_output_list_overflow(net_model_path):
    """The launch's list of flagged outputs capped far below what the batch flags (round-4 advisor finding: a tile whose
    reservation failed used to leave reserved-but-unwritten entries that the fix kernel walked): tiles that do not fit take
    the whole-tile recomputation; every byte still equals the oracle's."""
    n = 1536
    x = F.synth_features(n, 432, seed=12)
    x[::5] = np.float32("nan")  # 20 % of the rows flag every output: far more than a capped list holds, spread over all tiles
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    dnn.setInputLayerListCap(3000)
    dnn.setInputLayerKernel(4)
    got, recomputed = dnn.layer0(x)
    with np.errstate(all="ignore"):
        _, wt = Oracle(net_model_path).calculate(x, taps=True)
    assert np.array_equal(got, wt["u8_acts"][0])
    assert recomputed > n * 2048 // 5
    dnn.setInputLayerListCap(0)
    got2, _ = dnn.layer0(x)
    assert np.array_equal(got2, got)
    dnn.delete()
