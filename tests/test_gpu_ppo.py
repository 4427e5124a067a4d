"""The role-split fused OUTPUT kernel (fdnn_ppo.hip): one wave of every SIMD runs a half tile's k-loop while its partner
stages that half's operands and runs the soft-max of the half before -- exp in place in the accumulation registers, the
row sums exchanged between the 32 node tiles under the k-loop, the scale on the way out.  CalculateOutput + SoftMax::apply,
dnn.cc:428-454, :534-544.  Every bit must equal what fdnn_gemm.hip's in-phase fused tiles write (pinned against the oracle by
test_gpu_production_shapes / test_gpu_parity), and, directly, the oracle's rows to 2e-6 on EVERY row of the 10 000-frame batch.
The result buffer is poisoned before every pass: a row the kernel does not write shows."""
import os

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

TIGHT = 2e-6  # |p - oracle| per element (the tolerance of test_gpu_production_shapes: exp2-based exp, fp32 sums in tree order)


@pytest.fixture()
def modes():
    yield
    api.set_ppo(-1)


def device_pass(dnn, x, ppo):
    """One device-resident pass into a NaN-filled buffer -> (rows, give-ups of this pass)."""
    import torch

    api.set_ppo(ppo)
    n = x.shape[0]
    xd = torch.from_numpy(x).cuda()
    out = torch.full((n, dnn.outputDimension()), float("nan"), dtype=torch.float32, device="cuda")
    g0 = dnn.fuseGiveups()
    dnn.calculate_device(xd.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy(), dnn.fuseGiveups() - g0


@pytest.mark.parametrize("n", [1, 320, 321, 2560, 4097, 8500, 10000, 20480 + 77])
def test_role_split_output_equals_the_in_phase_fused_tiles(net_model_path, modes, n):
    """Sizes: one frame, one pair of halves, one frame more, exactly one pair per slot, workgroups with different numbers
    of pairs (4 097 = 13 pairs on 8 slots), an odd number of halves (8 500 = 53.1), the production batch (32 pairs: four
    per workgroup, the steady state), nine pairs per workgroup.  Twice, so that the counters the first pass leaves are used."""
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    x = F.synth_features(n, 432, seed=700 + n % 89)
    b, gb = device_pass(dnn, x, 1)
    a, ga = device_pass(dnn, x, 0)
    b2, gb2 = device_pass(dnn, x, 1)
    assert ga == 0 and gb == 0 and gb2 == 0
    assert not np.isnan(b).any()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b2.view(np.uint32))
    dnn.delete()


def test_role_split_output_every_row_against_the_oracle(net_model_path, modes):
    """configs[2]'s batch: all 10 000 rows of probabilities against CalculateOutput + SoftMax::apply (the oracle from every
    core) to 2e-6, every row's sum 1."""
    n = 10000
    x = F.synth_features(n, 432, seed=21)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got, gave_up = device_pass(dnn, x, 1)
    assert gave_up == 0
    orc = Oracle(net_model_path)
    want = orc.output_mt(orc.hidden_acts_mt(x))
    assert np.abs(got - want).max() <= TIGHT
    assert np.abs(got.sum(1, dtype=np.float64) - 1).max() < 1e-4
    dnn.delete()


def test_role_split_output_on_a_layer_without_saturating_pairs(modes):
    """The instance without the walk (trained, heavy-tailed nets have no risky pairs)."""
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_nosat.bin")
    F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="nosat")
    assert api.HostModel(p).risky_pairs(7) == 0
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 6000
    x = F.synth_features(n, 432, seed=3)
    b, gb = device_pass(dnn, x, 1)
    a, ga = device_pass(dnn, x, 0)
    assert ga == 0 and gb == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    orc = Oracle(p)
    want = orc.output_mt(orc.hidden_acts_mt(x))
    assert np.abs(b - want).max() <= TIGHT
    dnn.delete()


def test_role_split_output_with_corrections_firing_in_every_k_step(tmp_models, modes):
    """The pmaddubsw corrections (dnn.cc:337-340) of the OUTPUT layer inside the compute role: output weights near +-127
    make thousands of its pairs listed ones, and with activations near 255 the exact correction (not only the screen) runs
    -- on accumulators the compiler does not know about."""
    net = F.synth_net(list(F.NET_TOPOLOGY), seed=17)
    rng = np.random.default_rng(5)
    w = net.layers[-1].weights
    w[:] = rng.normal(0, 0.02, size=w.shape).astype(np.float32)
    hot = rng.random(w.shape) < 0.004
    w[hot] = rng.choice(np.array([-0.5, 0.5, 0.45, -0.48], np.float32), size=int(hot.sum()))
    p = os.path.join(tmp_models, "ppo_hot.bin")
    F.write_model_bin(p, net)
    n_q = len(F.NET_TOPOLOGY) - 2
    assert api.HostModel(p).risky_pairs(n_q) > 2048
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 700
    x = F.synth_features(n, 432, seed=8)
    b, gb = device_pass(dnn, x, 1)
    a, ga = device_pass(dnn, x, 0)
    assert ga == 0 and gb == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    orc = Oracle(p)
    hid = orc.hidden_acts_mt(x)
    want, acc = orc.output_mt(hid, want_acc=True)
    assert np.abs(b - want).max() <= TIGHT
    # the corrections did run: the accumulators differ from the plain (unsaturated) sums somewhere
    acc_dev, _ = dnn.productionOutputAcc(x, 1, probs=True)
    assert np.array_equal(acc_dev, acc)
    dnn.delete()


def test_the_switch_and_the_default(net_model_path, modes):
    """fdnn_debug_set_ppo rejects values outside {-1, 0, 1}.  The default (-1) takes the role-split kernel for a layer with
    saturating pairs from 22 frame pairs whose last round is 4/5 full (pair-free layers: 14 pairs, 3/4) -- whichever it takes, the bits are the same."""
    with pytest.raises(Exception):
        api.set_ppo(2)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    for n in (6000, 7040, 8000, 12345):
        x = F.synth_features(n, 432, seed=900 + n % 7)
        d, gd = device_pass(dnn, x, -1)
        a, ga = device_pass(dnn, x, 0)
        assert gd == 0 and ga == 0 and np.array_equal(a.view(np.uint32), d.view(np.uint32))
    dnn.delete()
