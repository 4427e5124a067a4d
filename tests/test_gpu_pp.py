"""The role-split hidden-layer kernel (fdnn_pp.hip): one wave of every SIMD runs a tile's k-loop while its partner stages
that tile's operands and runs the epilogue of the tile before -- QuantizedLayerActivations / quantizedNodeSum + AddBias +
QuantizedSigmoid, dnn.cc:250-349.  Every byte must equal what the in-phase tiles of fdnn_gemm.hip write (which
test_gpu_parity / test_gpu_production_shapes pin against the oracle), and, directly, what the oracle computes: on EVERY row of
the 10 000-frame batch."""
import os

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def modes():
    api.set_chain(0)  # one launch per layer: the path this kernel is a shape of
    yield
    api.set_chain(-1)
    api.set_pp(-1)


def hidden_bytes(dnn, x, pp):
    api.set_pp(pp, 1)
    ctx = dnn.getNewLazyContext(x.shape[0])
    ctx.calculateUntilOutput(x)
    got = ctx.hiddenActivations().copy()
    ctx.delete()
    return got


@pytest.mark.parametrize("n", [1, 320, 321, 4097, 8500, 10000, 12345, 20480 + 77])
def test_role_split_layers_equal_the_in_phase_tiles(net_model_path, modes, n):
    """Full 432 -> 7 x 2048 -> 8000 net (its layers have saturating pairs: the walk runs inside the compute role).  Sizes: one
    frame, one tile, one frame more, an odd number of half tiles (8 500 = 53.1 halves), the production batch, several tiles
    per workgroup (20 557 frames = 65 pairs x 8 node tiles on 256 workgroups: the steady state of the alternation)."""
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    x = F.synth_features(n, 432, seed=500 + n % 89)
    a = hidden_bytes(dnn, x, 0)
    b = hidden_bytes(dnn, x, 1)
    assert a.shape == (n, 2048) and np.array_equal(a, b)
    dnn.delete()


def test_role_split_layers_every_row_against_the_oracle(net_model_path, modes):
    """configs[2]'s batch: the last hidden layer's u8 activations of all 10 000 frames, bit for bit (the oracle from every core)."""
    n = 10000
    x = F.synth_features(n, 432, seed=21)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got = hidden_bytes(dnn, x, 1)
    want = Oracle(net_model_path).hidden_acts_mt(x)
    assert np.array_equal(got, want)
    dnn.delete()


def test_role_split_on_layers_without_saturating_pairs(tmp_models, modes):
    """The instance without the walk (trained, heavy-tailed nets have no risky pairs): a pair-free net of production width."""
    p = os.path.join(tmp_models, "pp_nosat.bin")
    F.write_model_bin(p, F.synth_net([432, 2048, 2048, 2048, 400], seed=41, mode="nosat"))
    assert api.HostModel(p).risky_pairs(1) == 0
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 5000
    x = F.synth_features(n, 432, seed=3)
    a = hidden_bytes(dnn, x, 0)
    b = hidden_bytes(dnn, x, 1)
    assert np.array_equal(a, b)
    assert np.array_equal(b, Oracle(p).hidden_acts_mt(x))
    dnn.delete()


def test_role_split_with_corrections_firing_in_every_k_step(tmp_models, modes):
    """The pmaddubsw corrections (dnn.cc:337-340) inside the compute role: weights near +-127 make thousands of pairs per
    64-node group listed ones, and with activations near 255 the exact correction (not only the screen) runs."""
    net = F.synth_net([432, 2048, 2048, 2048, 300], seed=17)
    rng = np.random.default_rng(5)
    for L in net.layers[1:3]:
        w = L.weights
        w[:] = rng.normal(0, 0.02, size=w.shape).astype(np.float32)
        hot = rng.random(w.shape) < 0.004  # ~4 listed pairs per node row of 1024 pairs
        w[hot] = rng.choice(np.array([-0.5, 0.5, 0.45, -0.48], np.float32), size=int(hot.sum()))
    p = os.path.join(tmp_models, "pp_hot.bin")
    F.write_model_bin(p, net)
    assert api.HostModel(p).risky_pairs(1) > 2048
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 700
    x = F.synth_features(n, 432, seed=8)
    a = hidden_bytes(dnn, x, 0)
    b = hidden_bytes(dnn, x, 1)
    assert np.array_equal(a, b)
    orc = Oracle(p)
    _, taps = orc.calculate(x[:32], taps=True)
    assert taps["sat_events"] > 0
    assert np.array_equal(b, orc.hidden_acts_mt(x))
    dnn.delete()
