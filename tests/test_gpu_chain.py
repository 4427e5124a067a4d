"""The int8 hidden layers as ONE persistent launch (fdnn_chain.hip) -- CalculateUntilLastHiddenLayer's layer loop,
dnn.cc:413-423, with a task per (layer, frame tile, node tile) drawn from per-XCD queues and a wait only for the task's
own frame tile in the layer before.  Every byte must equal what one launch per layer writes, which test_gpu_parity /
test_gpu_production_shapes pin against the oracle; here also directly against the oracle on sampled frames."""
import threading

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chain_mode():
    yield
    api.set_chain(-1)


def hidden_bytes(dnn, x, mode, min_frames=1):
    api.set_chain(mode, min_frames)
    ctx = dnn.getNewLazyContext(x.shape[0])
    ctx.calculateUntilOutput(x)
    got = ctx.hiddenActivations().copy()
    ctx.delete()
    return got


@pytest.mark.parametrize("n", [4097, 5000, 9000, 10000, 10241, 12000, 20480 + 77])
def test_chained_hidden_layers_equal_one_launch_per_layer(net_model_path, chain_mode, n):
    """Full 432 -> 7 x 2048 -> 8000 net, batch sizes on both sides of whole rounds of workgroups (a partial last round
    flows into the next layer here): the last hidden layer's u8 bytes, all of them, from both forms."""
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    x = F.synth_features(n, 432, seed=300 + n % 97)
    a = hidden_bytes(dnn, x, 0)
    b = hidden_bytes(dnn, x, 1)
    assert a.shape == (n, 2048) and np.array_equal(a, b)
    assert dnn.deviceCounters(8)[3] == 0  # no wait ran into its bound
    dnn.delete()


def test_chained_hidden_layers_against_the_oracle(net_model_path, chain_mode):
    """12 000 frames (38 frame tiles of 320: 1.19 rounds of workgroups per layer -- the chain's own case): 3 frames of
    every frame tile against the oracle's last hidden layer, bit for bit."""
    n = 12000
    x = F.synth_features(n, 432, seed=77)
    rng = np.random.default_rng(5)
    idx = np.array(sorted({min(n - 1, int(t0 + d)) for t0 in range(0, n, 320) for d in (0, int(rng.integers(1, 319)), 319)}))
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got = hidden_bytes(dnn, x, 1)[idx]
    _, taps = Oracle(net_model_path).calculate(x[idx], taps=True)
    assert np.array_equal(got, taps["u8_acts"][-1])
    dnn.delete()


def test_chain_on_a_net_with_every_pair_saturating(tmp_models, chain_mode):
    """The pmaddubsw corrections (dnn.cc:337-340) inside the chained kernel: weights at +-127 make every adjacent pair a
    listed one, so every k-step of every task runs the screen and most run the exact correction."""
    import os

    net = F.synth_net([432, 256, 256, 256, 256, 300], seed=17)
    rng = np.random.default_rng(5)
    for L in net.layers[1:]:
        L.weights[:] = rng.choice(np.array([-0.5, 0.5, 0.45, -0.48], np.float32), size=L.weights.shape)
    p = os.path.join(tmp_models, "chain_allsat.bin")
    F.write_model_bin(p, net)
    assert api.HostModel(p).risky_pairs(1) > 256 * 128 // 3
    dnn = api.QuantizedDnn.loadFromFile(p)
    n = 1500
    x = F.synth_features(n, 432, seed=8)
    a = hidden_bytes(dnn, x, 0)
    b = hidden_bytes(dnn, x, 1)
    assert np.array_equal(a, b)
    _, taps = Oracle(p).calculate(x[:64], taps=True)
    assert np.array_equal(b[:64], taps["u8_acts"][-1])
    dnn.delete()


def test_chain_repeated_and_from_many_streams(net_model_path, chain_mode):
    """The queue heads and per-frame-tile counters are rewound by the launch itself (last workgroup out, last task of a
    frame tile): 40 back-to-back passes on one context, then 6 threads with a context and a stream each, 12 000 frames --
    several chained launches share the chip, a workgroup of one may wait while the CUs run another's (no co-residency is
    assumed: whoever holds a task is running, and a task only waits for tasks drawn before it).  Bit-identical throughout."""
    import torch

    n = 12000
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    x = F.synth_features(n, 432, seed=4)
    want = hidden_bytes(dnn, x, 0)
    api.set_chain(1, 1)
    dx = torch.from_numpy(x).cuda()
    ctx = dnn.getNewLazyContext(n)
    for _ in range(40):
        ctx.calculateUntilOutputDevice(dx.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(ctx.hiddenActivations(), want)
    ctx.delete()
    bad = []

    def worker(t):
        s = torch.cuda.Stream()
        c = dnn.getNewLazyContext(n)
        for _ in range(10):
            c.calculateUntilOutputDevice(dx.data_ptr(), s.cuda_stream)
        s.synchronize()
        if not np.array_equal(c.hiddenActivations(), want):
            bad.append(t)
        c.delete()

    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not bad
    assert dnn.deviceCounters(8)[3] == 0
    dnn.delete()


def test_default_rule_picks_the_chain_only_where_it_saves_task_times(net_model_path, chain_mode):
    """Whole dense call under the default selection rule at a size the chain serves (12 000 frames) and one it does not
    (10 000): probabilities equal the one-launch-per-layer ones bit for bit."""
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    for n in (10000, 12000):
        x = F.synth_features(n, 432, seed=n)
        api.set_chain(0)
        a = dnn.calculate(x)
        api.set_chain(-1)
        b = dnn.calculate(x)
        assert np.array_equal(a, b)
    dnn.delete()
