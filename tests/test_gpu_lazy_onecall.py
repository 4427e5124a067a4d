"""One-call lazy scoring (fdnn_calculate_lazy / _bits / _bits_device): CalculateUntilLastHiddenLayer (dnn.cc:402-424) +
LazyOutputActivations (dnn.cc:355-392) for every frame of the call -- what LazyContext does in two calls per utterance
(QuantizedDnn.java:72-107) -- against the oracle's lazy path and against the two-call protocol, bit for bit."""
import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = 2e-6


@pytest.mark.parametrize("n", [1, 8, 100, 700])
def test_one_call_lazy_against_the_oracle_and_the_context_protocol(mid_model_path, n):
    x = F.synth_features(n, 432, seed=60 + n)
    O = 1000
    masks = F.generate_masks(n, O, 0.4, 0.03, seed=n)
    masks[0] = 0
    if n > 1:
        masks[1] = 1
    want = Oracle(mid_model_path).lazy(x, masks)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    got = dnn.calculateLazy(x, masks=masks)
    assert np.abs(got - want).max() <= TIGHT
    got_bits = dnn.calculateLazy(x, bits=F.pack_mask_bits(masks))
    assert np.array_equal(got, got_bits)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    two = ctx.calculateForOutputNodesBatch(masks)
    ctx.delete()
    assert np.array_equal(got, two)
    assert dnn.calculateLazy(np.zeros((0, 432), np.float32), masks=np.zeros((0, O), np.int8)).shape == (0, O)
    with pytest.raises(api.FdnnError):
        api._check(api.lib().fdnn_calculate_lazy(dnn.nativeDnnHandle, x.ctypes.data_as(api._c_f32p), n, 428, masks.ctypes.data_as(api._c_i8p),
                                                 got.ctypes.data_as(api._c_f32p)))
    dnn.delete()


def test_one_call_lazy_full_net_large_batch_and_device_form(net_model_path):
    """The full net at 10 000 frames (fused masked output kernel, bit masks read as they are): the host form's compacted
    return equals the device form's rows bit for bit, rows sum to one, 16 sampled frames equal the oracle's lazy rows."""
    import torch

    n, O = 10000, 8000
    x = F.synth_features(n, 432, seed=71)
    masks = F.generate_masks_fast(n, O, 0.40, 0.03, seed=3)
    bits = F.pack_mask_bits(masks)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got = dnn.calculateLazy(x, bits=bits)
    dx, db = torch.from_numpy(x).cuda(), torch.from_numpy(bits.view(np.int64)).cuda()
    od = torch.zeros((n, O), dtype=torch.float32, device="cuda")
    dnn.calculate_lazy_bits_device(dx.data_ptr(), n, db.data_ptr(), od.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(od.cpu().numpy(), got)
    assert np.abs(got.sum(1) - 1.0).max() < 1e-4
    idx = np.linspace(0, n - 1, 16).astype(int)
    want = Oracle(net_model_path).lazy(x[idx], masks[idx])
    assert np.abs(got[idx] - want).max() <= TIGHT
    dnn.delete()


def test_one_call_lazy_from_many_threads(mid_model_path):
    """The reference's concurrency model (MultiThreadedStressTest.java:48-61) on the one-call entry point: 8 threads, own
    utterances and masks, pooled contexts."""
    import threading

    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    orc = Oracle(mid_model_path)
    work = []
    for t in range(8):
        n = 50 + 13 * t
        x = F.synth_features(n, 432, seed=200 + t)
        m = F.generate_masks(n, 1000, 0.4, 0.03, seed=t)
        work.append((x, m, orc.lazy(x, m)))
    bad = []

    def run(t):
        x, m, want = work[t]
        for _ in range(10):
            if np.abs(dnn.calculateLazy(x, masks=m) - want).max() > TIGHT:
                bad.append(t)

    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not bad
    dnn.delete()
