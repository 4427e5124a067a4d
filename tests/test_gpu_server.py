"""The multi-stream scoring loop (fdnn_server_*): batches in flight and
coalesced host submissions must give exactly what the per-call API gives -- the serving shape of
QuantizedDnn.java:72-107 / MultiThreadedStressTest.java:48-69 on one GPU."""
import threading

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = 2e-6


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_device_batches_in_flight_are_deterministic(net_model_path, depth):
    """Full-size batches through the compute/tail stream pair: whatever the in-flight depth, every
    batch is bit-identical to fdnn_calculate_device on one stream (the soft-max scale running
    under the next batch's layer 0 touches only its own slot's scratch)."""
    import torch

    n, O = 6000, 8000
    xs = [torch.from_numpy(F.synth_features(n, 432, seed=70 + i)).cuda() for i in range(3)]
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    want = []
    for x in xs:
        o = torch.empty((n, O), dtype=torch.float32, device="cuda")
        dnn.calculate_device(x.data_ptr(), n, o.data_ptr(), torch.cuda.current_stream().cuda_stream)
        want.append(o)
    torch.cuda.synchronize()
    srv = api.ScoringServer(dnn, max_frames=n, depth=depth)
    outs = [torch.zeros((n, O), dtype=torch.float32, device="cuda") for _ in range(9)]
    torch.cuda.synchronize()
    tickets = [srv.submit_device(xs[i % 3].data_ptr(), n, outs[i].data_ptr()) for i in range(9)]
    for t in reversed(tickets):   # any order
        srv.wait(t)
    for i in range(9):
        assert torch.equal(outs[i], want[i % 3]), (depth, i)
    # lazy contract through the same loop
    masks = torch.from_numpy(F.generate_masks(300, O, 0.4, 0.03, seed=2)).cuda()
    ctx = dnn.getNewLazyContext(300)
    ctx.calculateUntilOutputDevice(xs[0].data_ptr(), 0)
    ref = torch.empty((300, O), dtype=torch.float32, device="cuda")
    ctx.calculateForOutputNodesBatchDevice(masks.data_ptr(), ref.data_ptr(), 0, 300, 0)
    torch.cuda.synchronize()
    ctx.delete()
    got = torch.zeros((300, O), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    srv.wait(srv.submit_device(xs[0].data_ptr(), 300, got.data_ptr(), masks.data_ptr()))
    assert torch.equal(got, ref)
    with pytest.raises(api.FdnnError):
        srv.submit_device(xs[0].data_ptr(), n + 1, outs[0].data_ptr())
    st = srv.stats()
    assert st["batches"] == 10 and st["frames"] == 9 * n + 300
    srv.close()
    dnn.delete()


def test_host_submissions_from_many_threads_are_coalesced(mid_model_path):
    """MultiThreadedStressTest.java:48-69 on the server: 8 threads x ragged utterances (some longer
    than one batch, some with lazy masks) -> every result bit-identical to the same utterance
    scored alone through QuantizedDnn.calculate / the lazy context, and equal to the oracle."""
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    orc = Oracle(mid_model_path)
    O = dnn.outputDimension()
    lens = [100, 37, 1, 250, 999, 100, 64, 1300, 100, 2, 511, 100]
    utts = [F.synth_features(n, 432, seed=400 + i) for i, n in enumerate(lens)]
    masks = {3: F.generate_masks(lens[3], O, 0.4, 0.03, seed=1), 7: F.generate_masks(lens[7], O, 0.4, 0.03, seed=2)}
    alone = []
    for i, x in enumerate(utts):
        if i in masks:
            ctx = dnn.getNewLazyContext(len(x))
            ctx.calculateUntilOutput(x)
            alone.append(ctx.calculateForOutputNodesBatch(masks[i]))
            ctx.delete()
        else:
            alone.append(dnn.calculate(x))
    assert np.abs(alone[0] - orc.calculate(utts[0])).max() <= TIGHT
    assert np.abs(alone[3] - orc.lazy(utts[3], masks[3])).max() <= TIGHT
    srv = api.ScoringServer(dnn, max_frames=1024, depth=3, linger_us=200)
    results = [None] * (8 * len(utts))
    errors = []

    def worker(w):
        try:
            for j in range(len(utts)):
                i = (j + w) % len(utts)
                t, out = srv.submit(utts[i], masks.get(i))
                srv.wait(t)
                results[w * len(utts) + j] = (i, out)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i, out in results:
        assert np.array_equal(out, alone[i]), i
    st = srv.stats()
    assert st["requests"] == 8 * len(utts)
    assert st["coalesced_requests"] > 0 and st["batches"] < st["requests"] + 8   # utterances really shared batches
    srv.close()
    dnn.delete()


def test_calculate_through_the_model_batcher(mid_model_path):
    """fdnn_model_enable_batcher: the unmodified calculate() entry (what the JNI symbol calls)
    from 8 threads goes through the coalescing loop and returns the same bits."""
    from concurrent.futures import ThreadPoolExecutor

    plain = api.QuantizedDnn.loadFromFile(mid_model_path)
    xs = [F.synth_features(20 + 17 * i, seed=300 + i) for i in range(12)]
    want = [plain.calculate(x) for x in xs]
    plain.delete()
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    dnn.enableBatcher(2048, 2, 100)
    with pytest.raises(api.FdnnError):
        dnn.enableBatcher(2048, 2, 100)
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda i: dnn.calculate(xs[i % 12]), range(96)))
    for i, g in enumerate(got):
        assert np.array_equal(g, want[i % 12]), i
    assert dnn.calculate(np.zeros((0, 432), np.float32)).shape == (0, 0)
    dnn.delete()


def test_server_argument_errors(tiny_model_path):
    dnn = api.QuantizedDnn.loadFromFile(tiny_model_path)
    with pytest.raises(api.FdnnError) as e:
        api.ScoringServer(dnn, 0, 2)
    assert e.value.code == api.FDNN_E_ARG
    with pytest.raises(api.FdnnError):
        api.ScoringServer(dnn, 100, 17)
    srv = api.ScoringServer(dnn, 64, 2)
    with pytest.raises(api.FdnnError) as e:
        srv.wait(12345)  # never issued
    assert e.value.code == api.FDNN_E_ARG
    with pytest.raises(ValueError):
        srv.submit(np.zeros((3, 429), np.float32))
    x = F.synth_features(10, 432, seed=1)
    with pytest.raises(ValueError):
        srv.submit(x, masks=np.ones((10, 99), np.int8))
    t, out = srv.submit(x)
    srv.wait(t)
    srv.wait(t)  # waiting twice is harmless
    assert np.array_equal(out, dnn.calculate(x))
    srv.drain()
    srv.close()
    dnn.delete()


def test_host_and_device_submissions_share_one_server(mid_model_path):
    """Both kinds of submission on the same slots at the same time: host utterances from worker
    threads (coalesced, staged through the slot's pinned buffers) while the main thread keeps
    device-resident batches in flight; every result equals the per-call API's."""
    import torch

    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    O = dnn.outputDimension()
    utts = [F.synth_features(50 + 30 * i, 432, seed=800 + i) for i in range(6)]
    want = [dnn.calculate(u) for u in utts]
    n = 3000  # above the small-batch threshold: compute stream + tail stream
    xd = torch.from_numpy(F.synth_features(n, 432, seed=9)).cuda()
    ref = torch.empty((n, O), dtype=torch.float32, device="cuda")
    dnn.calculate_device(xd.data_ptr(), n, ref.data_ptr(), 0)
    torch.cuda.synchronize()
    srv = api.ScoringServer(dnn, n, 2, 50)
    errors, got = [], {}

    def host_worker(w):
        try:
            for j in range(20):
                i = (j + w) % len(utts)
                t, out = srv.submit(utts[i])
                srv.wait(t)
                got[(w, j)] = (i, out)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=host_worker, args=(w,)) for w in range(4)]
    for t in th:
        t.start()
    outs = [torch.zeros((n, O), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(30):
        srv.wait(srv.submit_device(xd.data_ptr(), n, outs[k % 2].data_ptr()))
        assert torch.equal(outs[k % 2], ref), k
    for t in th:
        t.join()
    assert not errors, errors
    for (w, j), (i, out) in got.items():
        assert np.array_equal(out, want[i]), (w, j)
    srv.close()
    dnn.delete()


def test_very_large_batches_run_in_chunks(mid_model_path):
    """Dense passes over more than 20 480 frames run as chunks of whole 10 240-frame rounds (fdnn::frame_chunks) --
    back to back in fdnn_calculate_device, overlapped through the tail stream in the scoring loop,
    with and without masks.  Frames around every chunk boundary against the oracle, and the three
    paths against each other bit for bit."""
    import torch

    n, O = 33000, 1000   # chunks 20480 + 12520
    x = F.synth_features(n, 432, seed=77)
    idx = np.array([0, 1, 10238, 10239, 10240, 10241, 20478, 20479, 20480, 20481, 32998, 32999])
    orc = Oracle(mid_model_path)
    want = orc.calculate(x[idx])
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    xd = torch.from_numpy(x).cuda()
    a = torch.zeros((n, O), dtype=torch.float32, device="cuda")
    dnn.calculate_device(xd.data_ptr(), n, a.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.abs(a[torch.from_numpy(idx).cuda()].cpu().numpy() - want).max() <= TIGHT
    assert float((a.sum(1) - 1).abs().max()) < 1e-4
    # reference: the same frames in pieces that are cut differently (10240 + 760 each)
    b = torch.zeros_like(a)
    for lo in range(0, n, 11000):
        hi = min(n, lo + 11000)
        dnn.calculate_device(xd[lo:hi].data_ptr(), hi - lo, b[lo:hi].data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    srv = api.ScoringServer(dnn, n, 2)
    c1, c2 = torch.zeros_like(a), torch.zeros_like(a)
    t1 = srv.submit_device(xd.data_ptr(), n, c1.data_ptr())
    t2 = srv.submit_device(xd.data_ptr(), n, c2.data_ptr())
    srv.wait(t1)
    srv.wait(t2)
    assert torch.equal(c1, a) and torch.equal(c2, a)
    masks = torch.from_numpy(F.generate_masks_fast(n, O, 0.4, 0.03, seed=3)).cuda()
    lz = torch.zeros_like(a)
    srv.wait(srv.submit_device(xd.data_ptr(), n, lz.data_ptr(), masks.data_ptr()))
    sel = torch.from_numpy(idx).cuda()
    want_lazy = orc.lazy(x[idx], masks[sel].cpu().numpy())
    assert np.abs(lz[sel].cpu().numpy() - want_lazy).max() <= TIGHT
    srv.close()
    dnn.delete()


def test_create_and_free_cycles_return_device_memory(mid_model_path):
    """Models, lazy contexts, scoring loops (device and host side) and device groups created and freed
    in a loop: device memory in use must come back to where it started (hipMemGetInfo through torch)."""
    import torch

    x = F.synth_features(700, 432, seed=9)

    def cycle():
        dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
        dnn.calculate(x)
        ctx = dnn.getNewLazyContext(700)
        ctx.calculateUntilOutput(x)
        ctx.calculateForOutputNodes(np.ones(dnn.outputDimension(), dtype=np.int8))
        ctx.delete()
        srv = api.ScoringServer(dnn, 2048, 3)
        t, _ = srv.submit(x)
        srv.wait(t)
        xd = torch.from_numpy(x).cuda()
        od = torch.empty((700, dnn.outputDimension()), dtype=torch.float32, device="cuda")
        srv.wait(srv.submit_device(xd.data_ptr(), 700, od.data_ptr()))
        srv.close()
        dnn.enableBatcher(1024, 2, 0)
        dnn.calculate(x[:100])
        dnn.delete()
        grp = api.DeviceGroup(mid_model_path, [0, 0])
        grp.calculate(x[:64])
        grp.delete()
        del xd, od

    for _ in range(3):
        cycle()  # pools, caches and the allocator's own slack settle
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _total = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _total = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB of device memory did not come back after 25 create/free cycles"


def test_lazy_bit_mask_submissions_are_coalesced_and_come_back_compacted(mid_model_path):
    """fdnn_server_submit_lazy_bits: LazyContext's contract (QuantizedDnn.java:72-107, dnn.cc:355-392) for many callers --
    8 threads x ragged utterances with bit masks, dense callers in between (they never share a bit-mask batch), an
    utterance longer than a batch (its pieces come back compacted one by one), a nearly-all-active mask (its batch
    leaves whole) and an all-inactive one.  Every row equals the same utterance through the one-call lazy entry bit for
    bit, and the oracle's lazy rows."""
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    orc = Oracle(mid_model_path)
    O = dnn.outputDimension()
    lens = [100, 37, 1, 250, 100, 1300, 64, 100, 2, 511]
    utts = [F.synth_features(n, 432, seed=500 + i) for i, n in enumerate(lens)]
    masks = [F.generate_masks(n, O, 0.4, 0.03, seed=20 + i) for i, n in enumerate(lens)]
    masks[2][:] = 0             # no active node at all
    masks[6][:] = 1
    masks[6][:, ::17] = 0       # ~94 % active: this batch is not worth compacting
    masks[4][3] = 1             # one all-active row inside an ordinary utterance
    bits = [F.pack_mask_bits(m) for m in masks]
    dense_ids = {7, 8}          # these two are submitted dense
    alone = [dnn.calculate(x) if i in dense_ids else dnn.calculateLazy(x, bits=bits[i]) for i, x in enumerate(utts)]
    for i in (0, 2, 5):
        assert np.abs(alone[i] - orc.lazy(utts[i], masks[i])).max() <= TIGHT
    srv = api.ScoringServer(dnn, max_frames=1024, depth=3, linger_us=200)
    results, errors = [], []
    lock = threading.Lock()

    def worker(w):
        try:
            for j in range(len(utts)):
                i = (j + w) % len(utts)
                out = np.full((lens[i], O), -7.0, dtype=np.float32)
                t, _ = srv.submit(utts[i], out=out) if i in dense_ids else srv.submitLazy(utts[i], bits[i], out=out)
                srv.wait(t)
                with lock:
                    results.append((i, out))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 8 * len(utts)
    for i, out in results:
        assert np.array_equal(out, alone[i]), i
    st = srv.stats()
    assert st["coalesced_requests"] > 0
    with pytest.raises(api.FdnnError):
        api._check(api.lib().fdnn_server_submit_lazy_bits(srv.handle, utts[0].ctypes.data_as(api._c_f32p), 100, None,
                                                          results[0][1].ctypes.data_as(api._c_f32p), None))
    srv.close()
    dnn.delete()


def test_one_call_lazy_through_the_model_batcher(mid_model_path):
    """fdnn_model_enable_batcher also carries the one-call lazy entry (the JNI extension calculateLazyBatch): 8 threads,
    lazy and dense calls mixed, same bits as without the batcher."""
    from concurrent.futures import ThreadPoolExecutor

    O = 1000
    xs = [F.synth_features(20 + 17 * i, seed=330 + i) for i in range(10)]
    ms = [F.generate_masks(len(x), O, 0.4, 0.03, seed=40 + i) for i, x in enumerate(xs)]
    plain = api.QuantizedDnn.loadFromFile(mid_model_path)
    want = [plain.calculateLazy(x, masks=m) if i % 3 else plain.calculate(x) for i, (x, m) in enumerate(zip(xs, ms))]
    plain.delete()
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    dnn.enableBatcher(2048, 2, 100)

    def one(k):
        i = k % 10
        return dnn.calculateLazy(xs[i], masks=ms[i]) if i % 3 else dnn.calculate(xs[i])

    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(one, range(80)))
    for k, g in enumerate(got):
        assert np.array_equal(g, want[k % 10]), k
    dnn.delete()


def test_lazy_loop_on_the_full_net_with_large_coalesced_batches(net_model_path):
    """BASELINE configs[1]-sized utterances (1000 frames) on the 432 -> 7x2048 -> 8000 net, eight callers, bit masks through
    the scoring loop: the coalesced batches (up to 6400 frames) take the large-batch kernels -- int8-screened layer 0, the
    masked output instance with the soft-max inside, the bits read as they are -- and come back compacted; every row equals
    the same utterance through the one-call entry, sampled rows equal the oracle's LazyOutputActivations (dnn.cc:355-392)."""
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    O = dnn.outputDimension()
    lens = [1000, 1000, 640, 1000, 100, 1000, 2700, 1000]
    utts = [F.synth_features(n, 432, seed=900 + i) for i, n in enumerate(lens)]
    masks = [F.generate_masks_fast(n, O, 0.40, 0.03, seed=60 + i) for i, n in enumerate(lens)]
    bits = [F.pack_mask_bits(m) for m in masks]
    alone = [dnn.calculateLazy(x, bits=b) for x, b in zip(utts, bits)]
    orc = Oracle(net_model_path)
    idx = np.array([0, 1, 499, 999])
    assert np.abs(alone[0][idx] - orc.lazy(utts[0][idx], masks[0][idx])).max() <= TIGHT
    srv = api.ScoringServer(dnn, max_frames=6400, depth=3, linger_us=300)
    outs = [np.full((n, O), -5.0, dtype=np.float32) for n in lens]
    errors = []

    def worker(i):
        try:
            for _ in range(3):
                outs[i].fill(-5.0)
                t, _ = srv.submitLazy(utts[i], bits[i], out=outs[i])
                srv.wait(t)
                assert np.array_equal(outs[i], alone[i]), i
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(lens))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    st = srv.stats()
    assert st["coalesced_requests"] > 0 and st["frames"] == 3 * sum(lens)
    assert dnn.fuseGiveups() == 0
    srv.close()
    dnn.delete()
