"""Parity of the kernel instances production actually dispatches at BASELINE's full sizes.

The small-batch tests of test_gpu_parity.py run the 32/64/128-frame 4-wave tiles.  A 10 000-frame
batch of the 7x2048 -> 8000 net runs the 8-wave 256-node x 320-frame tile (rotated-barrier k-loop,
16 k-steps, saturation walk, mask staged through LDS): these tests meet the oracle THERE --
every row of the 10 000-frame batches, integer state bit for bit (the oracle threaded over every core) -- plus the
size-independent properties on every row (reference: dnn.cc:355-392, :402-454)."""
import os

import numpy as np
import pytest

from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = 2e-6


def sample_every_tile(n, tile, per_tile, seed):
    """`per_tile` frame indices out of every `tile`-frame tile of [0, n): first and last row of the
    tile (the wave / MFMA-block edges) and random rows in between."""
    rng = np.random.default_rng(seed)
    idx = []
    for t0 in range(0, n, tile):
        hi = min(t0 + tile, n)
        pick = {t0, hi - 1}
        while len(pick) < min(per_tile, hi - t0):
            pick.add(int(rng.integers(t0, hi)))
        idx.extend(sorted(pick))
    return np.array(idx)


_ORACLE_CACHE = {}


@pytest.fixture(params=["default", "scale-pass"])
def softmax_path(request):
    """Large batches under both soft-max arrangements: the default (scaled inside the output kernel) and the separate scale
    pass -- what FDNN_FUSE_NORM=0, or a second process on the GPU, selects (a hand-kept log of such runs until round 4)."""
    api.set_fuse(0 if request.param == "scale-pass" else -1)
    yield request.param
    api.set_fuse(-1)


def _oracle_dense10k(net_model_path, x):
    """The oracle over EVERY row of the batch, from every core (ctypes drops the GIL): last hidden layer, production
    accumulators of the output layer, probabilities -- about a second per 10 000 frames on 16 cores."""
    if "dense10k" not in _ORACLE_CACHE:
        orc = Oracle(net_model_path)
        hid = orc.hidden_acts_mt(x)
        probs, acc = orc.output_mt(hid, want_acc=True)
        _, taps = orc.calculate(x[:8], taps=True)  # (the gauss net does saturate: the fix-up walk is exercised)
        _ORACLE_CACHE["dense10k"] = (hid, probs, acc, taps["sat_events"])
    return _ORACLE_CACHE["dense10k"]


def test_dense_10k_every_row_against_the_oracle(net_model_path, softmax_path):
    """configs[2], all 10 000 rows (round 5 compared 512 of them): the last hidden layer's u8 activations (six 16-step
    rotated k-loops with the saturation walk behind them) and the PRODUCTION output instance's int32 accumulators bit for
    bit, the soft-max rows to 2e-6 (dnn.cc:402-454)."""
    n = 10000
    x = F.synth_features(n, 432, seed=21)
    want_hid, want_p, want_acc, sat = _oracle_dense10k(net_model_path, x)
    assert sat > 0
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    hid = ctx.hiddenActivations()
    ctx.delete()
    assert np.array_equal(hid, want_hid)
    acc, p = dnn.productionOutputAcc(x, 1, probs=True)
    assert np.array_equal(acc, want_acc)
    assert np.abs(p - want_p).max() <= TIGHT
    p2 = dnn.calculate(x)
    assert np.abs(p2 - want_p).max() <= TIGHT
    assert np.abs(p2.sum(1, dtype=np.float64) - 1).max() < 1e-4
    dnn.delete()


def test_lazy_10k_masked_kernel_every_row_against_the_oracle(net_model_path, softmax_path):
    """configs[3]: 10 000 frames, 40 % mask with 3 % churn (FuncTest.java:121-133) through the
    device-pointer batched lazy call = qgemm_kernel<5,2,128,2,OUTPUT,..,PLAIN,MASKED>.  EVERY row against
    LazyOutputActivations (dnn.cc:355-392) to 2e-6 (round 5: 512 rows); on every row also:
    sum == 1, masked-out nodes all equal 1/total (exp(0) terms, dnn.cc:366-369)."""
    import torch

    n, O = 10000, 8000
    x = F.synth_features(n, 432, seed=51)
    masks = F.generate_masks(n, O, 0.40, 0.03, seed=7)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(n)
    xd = torch.from_numpy(x).cuda()
    md = torch.from_numpy(masks).cuda()
    od = torch.zeros((n, O), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.calculateUntilOutputDevice(xd.data_ptr(), s)
    ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, s)
    torch.cuda.synchronize()
    # properties on all rows, on the device (320 MB)
    assert float((od.sum(1, dtype=torch.float64) - 1).abs().max()) < 1e-4
    off = md == 0
    lo = torch.where(off, od, torch.full_like(od, float("inf"))).min(1).values
    hi = torch.where(off, od, torch.full_like(od, float("-inf"))).max(1).values
    assert bool((lo == hi).all()) and float(lo.min()) > 0   # one value per row for the masked-out nodes
    got = od.cpu().numpy()
    if "lazy10k" not in _ORACLE_CACHE:
        orc = Oracle(net_model_path)
        _ORACLE_CACHE["lazy10k"] = orc.output_mt(orc.hidden_acts_mt(x), masks=masks)
    want = _ORACLE_CACHE["lazy10k"]
    assert np.abs(got - want).max() <= TIGHT
    # all-ones masks through the masked instance == the dense instance, bit for bit
    md.fill_(1)
    ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, s)
    dense = torch.empty_like(od)
    dnn.calculate_device(xd.data_ptr(), n, dense.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(od, dense)
    ctx.delete()
    dnn.delete()


@pytest.mark.parametrize("out_dim,n", [(1003, 12000), (1000, 20000), (1003, 20000)])
def test_small_net_big_batch_masked_8_wave_tiles(tmp_models, out_dim, n):
    """A net small enough for the oracle to score EVERY frame, with enough frames that
    rows_pad/256 * ceil(n/128) > 256 picks the 8-wave 256/320-frame tiles: the MASKED (+ANYW for a
    width that is not a multiple of four) epilogue with its LDS-staged mask, all frames.
    12 000 frames -> 256-frame tiles, 20 000 -> 320-frame tiles (qgemm_frame_tile's cost model)."""
    p = os.path.join(tmp_models, f"smallnet_out{out_dim}.bin")
    F.write_model_bin(p, F.synth_net([432, 128, 128, 128, out_dim], seed=90 + out_dim))
    x = F.synth_features(n, 432, seed=out_dim)
    masks = F.generate_masks(n, out_dim, 0.40, 0.03, seed=out_dim + 3)
    orc = Oracle(p)
    want_lazy = orc.lazy(x, masks)
    want_dense, wt = orc.calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    assert (ctx.hiddenActivations() == wt["u8_acts"][-1]).all()
    got = ctx.calculateForOutputNodesBatch(masks)
    assert np.abs(got - want_lazy).max() <= TIGHT
    sub = ctx.calculateForOutputNodesBatch(masks[777:11000], first=777)  # unaligned first row, ragged last tile
    assert np.array_equal(sub, got[777:11000])
    ctx.delete()
    assert np.abs(dnn.calculate(x) - want_dense).max() <= TIGHT
    dnn.delete()


def test_context_orders_device_and_host_entry_points(net_model_path):
    """A context's *_device entries run on the caller's stream, its host-pointer entries on its
    own non-blocking stream.  The library orders them (fdnn.h, "stream ordering on a context"):
    hidden layers enqueued on a side stream -- behind a long kernel, so they are certainly not
    finished when the next call is made -- followed at once by host-pointer reads."""
    import torch

    n = 2000
    x = F.synth_features(n, 432, seed=61)
    masks = F.generate_masks(8, 8000, 0.40, 0.03, seed=3)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ref = dnn.getNewLazyContext(n)
    ref.calculateUntilOutput(x)
    want_hidden = ref.hiddenActivations()
    want_rows = np.stack([ref.calculateForOutputNodes(masks[i]) for i in range(8)])
    ref.delete()
    side = torch.cuda.Stream()
    xd = torch.from_numpy(x).cuda()
    big = torch.randn((8192, 8192), device="cuda")
    torch.cuda.synchronize()
    for trial in range(3):
        ctx = dnn.getNewLazyContext(n)
        with torch.cuda.stream(side):
            for _ in range(4):
                big = (big @ big).clamp_(-1, 1)          # ~10 ms of work ahead of the hidden layers
        ctx.calculateUntilOutputDevice(xd.data_ptr(), side.cuda_stream)
        if trial == 0:
            got = ctx.hiddenActivations()                 # host entry, context stream: no sync by the caller
            assert (got == want_hidden).all()
        rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(8)])
        assert (rows == want_rows).all()
        # and back: host-pointer forward, then a device-pointer output on the side stream
        ctx.calculateUntilOutput(x)
        od = torch.zeros((8, 8000), dtype=torch.float32, device="cuda")
        md = torch.from_numpy(masks).cuda()
        ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, 8, side.cuda_stream)
        side.synchronize()
        assert (od.cpu().numpy() == want_rows).all()
        ctx.delete()
    dnn.delete()


def test_net_with_every_pair_saturating(tmp_models):
    """Weights near +-127 everywhere: nearly every adjacent pair can leave int16, so the sparse
    correction walks tens of thousands of entries per 64-node group and pmaddubsw really
    saturates for most of them (dnn.cc:337-340).  Slow by design, bit-exact all the same."""
    net = F.synth_net([432, 256, 256, 256, 300], seed=17)
    rng = np.random.default_rng(5)
    for L in net.layers[1:]:
        L.weights[:] = rng.choice(np.array([-0.5, 0.5, 0.45, -0.48], np.float32), size=L.weights.shape)
    p = os.path.join(tmp_models, "allsat_small.bin")
    F.write_model_bin(p, net)
    hm = api.HostModel(p)
    assert hm.risky_pairs(1) > 256 * 128 // 3
    x = F.synth_features(700, 432, seed=9)
    want, wt = Oracle(p).calculate(x, taps=True)
    assert wt["sat_events"] > 100000
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["acc_hid"] == wt["acc_hid"]).all() and (t["acc_out"] == wt["acc_out"]).all()
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    assert np.abs(dnn.calculate(x) - want).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("exact_path", [False, True])
def test_sigmoid_table_ends(tmp_models, exact_path):
    """Activations far into both tails of the sigmoid (|x| > 6.4 -> table entries 0 and 255,
    dnn.h:38-41) in the int8 layers: the last entries of the half-step table (fast epilogue) and
    of the 1281-entry table (exact epilogue, forced by one bias that breaks the int32 bound) sit
    in the final 16-byte piece of their LDS copy."""
    net = F.synth_net([432, 128, 128, 128, 200], seed=31)
    net.layers[1].weights[:] *= 8.0
    net.layers[2].weights[:] *= 8.0
    net.layers[1].bias[1::5] = 9.0      # k >= 640 -> 255
    net.layers[1].bias[2::5] = -9.0     # k <= -640 -> 0
    if exact_path:
        net.layers[1].bias[0] = 3.0e7   # |lin| * 200 no longer provably < 2^31 -> exact round()/table path
        net.layers[2].bias[0] = -3.0e7
    p = os.path.join(tmp_models, f"tails_{int(exact_path)}.bin")
    F.write_model_bin(p, net)
    x = F.synth_features(500, 432, seed=4)
    want, wt = Oracle(p).calculate(x, taps=True)
    u = wt["u8_acts"]
    assert (u[1] == 255).mean() > 0.15 and (u[1] == 0).mean() > 0.15
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == u).all()
    assert (t["acc_out"] == wt["acc_out"]).all()
    ctx = dnn.getNewLazyContext(500)      # production (tap-free) instances
    ctx.calculateUntilOutput(x)
    assert (ctx.hiddenActivations() == u[-1]).all()
    ctx.delete()
    dnn.delete()


def test_layer0_128_node_tile_still_bit_exact():
    """The chain kernel ships on 64-node tiles (partial sums in registers); the round-1 shape --
    128-node tiles with l2 + l3 parked in global scratch -- stays selectable (FDNN_L0_TN=128, read
    once per process): run the layer-0 parity tests under it in a child process."""
    import subprocess
    import sys

    env = dict(os.environ, FDNN_L0_TN="128")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "input_widths or tiny_golden or blob_export or ragged_batch"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("seed,scale", [(3, 1.0), (4, 4.0)])
def test_screened_layer0_is_bit_exact(net_model_path, tmp_models, seed, scale):
    """Large batches compute layer 0 with FUSED chains on the fp32 matrix pipe and recompute, with the
    exact unfused chains (dnn.cc:219-247), only the outputs whose table index the fusion could change
    (fdnn_l0.hip, "screened").  Every one of 4096 x 2048 layer-0 activations of the production kernels
    against the oracle's canonical layer 0, bit for bit -- also with inputs four times larger (more
    outputs in the sigmoid's tails, wider error bounds)."""
    n = 4096
    x = (F.synth_features(n, 432, seed=seed) * np.float32(scale)).astype(np.float32)
    want, wt = Oracle(net_model_path).calculate(x[:8], taps=True)   # cheap sanity that the oracle is up
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got, recomputed = dnn.layer0(x)
    import os

    if os.environ.get("FDNN_L0_NO_SCREEN"):  # (diagnosis switch: the all-VALU chain kernel instead; nothing is screened)
        assert recomputed == 0
    else:
        assert 0 < recomputed < 0.25 * n * 2048, recomputed   # the screened path ran, and screening is selective
    # the oracle's canonical layer 0 for all frames: hidden layers are not needed, so score in slices
    orc = Oracle(net_model_path)
    for lo in range(0, n, 512):
        _, t = orc.calculate(x[lo:lo + 512], taps=True)
        bad = np.argwhere(got[lo:lo + 512] != t["u8_acts"][0])
        assert bad.size == 0, (lo, bad[:5].tolist())
    # the all-VALU chain kernel gives the same bytes (kind 1 forces it)
    dnn.setInputLayerKernel(1)
    chain, rec2 = dnn.layer0(x)
    assert rec2 == 0 and np.array_equal(chain, got)
    dnn.delete()


def test_screened_layer0_with_a_wide_input_is_bit_exact(tmp_models):
    """The screened path's error bound carries a term that grows with the square of the input width
    ((D^2/2 + 2D) u^2 S: the unfused chain's own partial sums against the sampled fused ones), which at the
    432-wide benchmark input hides inside the bound's slack and at 2048 does not.  A 2048-wide input layer,
    2304 frames (screened path: large batch, no taps), every activation byte against the oracle."""
    p = os.path.join(tmp_models, "wide_in.bin")
    F.write_model_bin(p, F.synth_net([2048, 256, 256, 256, 64], seed=21, w0_std=0.01))
    n = 2304
    x = F.synth_features(n, 2048, seed=9, pad_from=None)
    dnn = api.QuantizedDnn.loadFromFile(p)
    dnn.setInputLayerKernel(3)  # on a layer this narrow the cost model would pick the chain kernel
    got, recomputed = dnn.layer0(x)
    if os.environ.get("FDNN_L0_NO_SCREEN"):  # (diagnosis switch: nothing is screened)
        assert recomputed == 0
    else:
        assert 0 < recomputed < 0.5 * n * 256, recomputed
    orc = Oracle(p)
    for lo in range(0, n, 768):
        _, t = orc.calculate(x[lo:lo + 768], taps=True)
        bad = np.argwhere(got[lo:lo + 768] != t["u8_acts"][0])
        assert bad.size == 0, (lo, bad[:5].tolist())
    dnn.delete()


@pytest.mark.parametrize("n,stride", [(10000, 41), (100, 7)])
def test_production_output_instance_accumulators_are_bit_exact(net_model_path, n, stride):
    """The int32 accumulators of the output layer, 8000 nodes wide, out of the PRODUCTION kernel instances (the
    branch-free dense instance of the 10 000-frame batch; the small-batch kernel at 100 frames) -- not the tap kernels,
    which are different instances: every stride-th frame's 8000 sums against the oracle's quantizedNodeSum values
    (dnn.cc:323-349, pair saturation included), bit for bit.  A wrong low bit of one accumulator would hide inside
    the 2e-6 the soft-max comparison allows."""
    x = F.synth_features(n, 432, seed=77)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    acc, probs = dnn.productionOutputAcc(x, stride, probs=True)
    plain = dnn.calculate(x)
    dnn.delete()
    assert np.array_equal(probs, plain)           # the probe does not change what the call returns
    idx = np.arange(0, n, stride)
    assert acc.shape == (len(idx), 8000)
    want, wt = Oracle(net_model_path).calculate(x[idx], taps=True)
    assert np.array_equal(acc, wt["acc_out"]), np.argwhere(acc != wt["acc_out"])[:5].tolist()
    assert np.abs(probs[idx] - want).max() <= 2e-6


@pytest.mark.parametrize("n", [2100, 3000, 4096])
def test_mid_size_batches_on_the_128_node_shape(net_model_path, n):
    """2 049 .. 4 096 frames on the 2048-node layers take the four-wave 128-node x 128-frame tile (two workgroups per CU:
    qgemm_kernel<2,2,128,2,..,WM=2>), the 8000-node output layer the 256-node tiles.  Eight frames of every 128-frame
    tile: last hidden layer bit for bit against the oracle, soft-max rows to 2e-6; the production output instance's
    int32 accumulators of every 64th frame bit for bit."""
    x = F.synth_features(n, 432, seed=300 + n)
    idx = sample_every_tile(n, 128, 8, seed=2)
    orc = Oracle(net_model_path)
    want, wt = orc.calculate(x[idx], taps=True)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    hid = ctx.hiddenActivations()
    ctx.delete()
    assert (hid[idx] == wt["u8_acts"][-1]).all()
    acc, p = dnn.productionOutputAcc(x, 64, probs=True)
    dnn.delete()
    assert np.abs(p[idx] - want).max() <= TIGHT
    _, wt64 = orc.calculate(x[::64], taps=True)
    assert np.array_equal(acc, wt64["acc_out"])


def test_fused_softmax_equals_the_scale_pass_and_its_give_up_path(net_model_path, tmp_path):
    """Large dense batches scale the soft-max inside the output kernel (fused: exp(z) stays in registers, the 256-node
    tiles of a frame tile exchange row sums through memory).  Same bits as the unfused kernel + normalize pass
    (FDNN_FUSE_NORM=0), and the same bits again when every third node tile pretends its wait timed out
    (FDNN_GEMM_DEBUG=4096): the in-phase tiles (the masked call here) store that part of exp(z) unscaled and their frame tile's
    last workgroup finishes them; the role-split dense kernel (fdnn_ppo.hip: the dense call here) leaves its half's rows
    unwritten, raises the model's fault word, and fdnn_calculate runs the output layer again, unfused.  Whichever call comes
    first in a process meets the give-up (after it the model does not fuse any more): both orders."""
    import subprocess, sys

    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from fast_dnn_amd import api, formats as F
x = F.synth_features(10000, 432, seed=41)
masks = F.generate_masks_fast(10000, 8000, 0.40, 0.03, seed=3)
dnn = api.QuantizedDnn.loadFromFile({net_model_path!r})
def dense():
    return dnn.calculate(x)
def lazy():
    ctx = dnn.getNewLazyContext(10000)
    ctx.calculateUntilOutput(x)
    q_ = ctx.calculateForOutputNodesBatch(masks)
    ctx.delete()
    return q_
if sys.argv[2] == "dense-first":
    p = dense(); q = lazy()
else:
    q = lazy(); p = dense()
print("GIVEUPS", dnn.fuseGiveups())
dnn.delete()
np.save(sys.argv[1], np.concatenate([p[::7], q[::7]]))
"""
    outs = []
    # (FDNN_FUSE_NORM=1: this pytest process may hold the device's advisory marker -- it loaded models in earlier tests --, and
    # a child that finds it taken would run unfused by itself: the fused variants must really fuse)
    for tag, env, order in (("fused", {"FDNN_FUSE_NORM": "1"}, "dense-first"), ("unfused", {"FDNN_FUSE_NORM": "0"}, "dense-first"),
                            ("giveup_dense", {"FDNN_FUSE_NORM": "1", "FDNN_GEMM_DEBUG": "4096"}, "dense-first"),
                            ("giveup_lazy", {"FDNN_FUSE_NORM": "1", "FDNN_GEMM_DEBUG": "4096"}, "lazy-first")):
        f = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, f, order], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        gave_up = int(r.stdout.split("GIVEUPS")[1].split()[0])
        assert (gave_up > 0) == tag.startswith("giveup"), (tag, gave_up)
        if tag.startswith("giveup"):
            assert r.stderr.count("sat out its bounded wait") == 1
        outs.append(np.load(f))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[3])
    assert np.abs(outs[0].sum(1, dtype=np.float64) - 1).max() < 1e-4


@pytest.mark.parametrize("kind", [4, 3])
def test_both_screening_kernels_are_bit_exact(net_model_path, kind):
    """Round 4 moved the screening of large-batch layer 0 from the fp32 matrix pipe (kind 3) to exact integer arithmetic
    on the int8 pipe (kind 4, fdnn_l0s.hip): 24-bit integer images of both operands, eight int8 MFMA products, sampled
    partial sums, a rigorous bound, exact recomputation of the flagged outputs (dnn.cc:219-247).  Both stay selectable;
    each against the oracle over 3000 x 2048 bytes at a ragged frame count."""
    n = 3000 + 37
    x = F.synth_features(n, 432, seed=12)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    dnn.setInputLayerKernel(kind)
    got, recomputed = dnn.layer0(x)
    screened = not os.environ.get("FDNN_L0_NO_SCREEN") and not (kind == 4 and os.environ.get("FDNN_L0_NO_SPLIT"))
    if screened:  # (diagnosis switches take the screening paths out: the bytes below must still be right)
        assert 0 < recomputed < 0.05 * n * 2048, recomputed
    orc = Oracle(net_model_path)
    for lo in range(0, n, 512):
        _, t = orc.calculate(x[lo:lo + 512], taps=True)
        bad = np.argwhere(got[lo:lo + 512] != t["u8_acts"][0])
        assert bad.size == 0, (kind, lo, bad[:5].tolist())
    dnn.delete()


def test_int8_screening_with_hostile_rows(net_model_path):
    """What the 24-bit integer image of a frame row cannot represent must fall to the exact path, never to a wrong byte:
    rows with one element 10^6 times the rest (every other element loses its digits), all-zero rows, a row of denormals,
    rows with inf / NaN (the reference's own result, whatever it is, bit for bit), huge and tiny uniform scales, and rows
    whose sums cancel to a few ulps.  1280 frames through the int8 screening (kind 4) against the oracle, every byte."""
    rng = np.random.default_rng(5)
    n = 1280
    x = F.synth_features(n, 432, seed=31)
    x[0:64, 7] = 1.0e6                                   # one dominant element
    x[64:96] = 0.0                                        # empty rows
    x[96:128] = (rng.standard_normal((32, 432)) * 1e-41).astype(np.float32)   # denormal inputs
    x[128:160] *= np.float32(1e-12)
    x[160:192] *= np.float32(1e12)
    x[192, 3] = np.inf
    x[193, 5] = -np.inf
    x[194, 11] = np.nan
    x[195] = np.float32(3.0e38)
    x[200:264, :216] = -x[200:264, 216:]                  # (shift/scale sit between this and the layer, so only near-cancelling)
    x[264:296] = np.round(x[264:296] * 4) / 4             # many exact ties in the products
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    dnn.setInputLayerKernel(4)
    got, recomputed = dnn.layer0(x)
    if not os.environ.get("FDNN_L0_NO_SCREEN") and not os.environ.get("FDNN_L0_NO_SPLIT"):
        assert recomputed > 0
    orc = Oracle(net_model_path)
    with np.errstate(all="ignore"):
        _, t = orc.calculate(x, taps=True)
    bad = np.argwhere(got != t["u8_acts"][0])
    assert bad.size == 0, bad[:8].tolist()
    dnn.setInputLayerKernel(1)   # and the all-VALU chain kernel agrees with both
    chain, _ = dnn.layer0(x)
    assert np.array_equal(chain, got)
    dnn.delete()


def test_fused_softmax_under_many_streams_and_two_models(net_model_path, tmp_models):
    """The fused soft-max's workgroups wait for their frame tile's other node tiles -- safe while one such kernel is being
    dispatched, a latency cliff (bounded waits of tens of milliseconds, then the clean-up kernel) when nine partially
    dispatched frame tiles from different launches fill the chip.  The library chains the fused launches of a device
    (fdnn_runtime.cpp: FuseChain), whatever stream, context or model they come from.  Here: 8 caller threads, each with
    its own stream, fdnn_calculate_device at 10 000 frames on the full net, and a ninth thread scoring a SECOND model on
    the same device, 50 rounds -- every result bit-identical to the single-stream one, no tile ever left to the
    clean-up kernel (fdnn_model_fuse_giveups), no call far beyond the median.  (SoftMax::apply, dnn.cc:534-544;
    concurrency model: MultiThreadedStressTest.java:48-61.)"""
    import threading
    import time

    import torch

    n, T, rounds = 10000, 8, 50
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    p2 = os.path.join(tmp_models, "second_fused.bin")
    F.write_model_bin(p2, F.synth_net([432, 512, 512, 512, 2048], seed=77))
    dnn2 = api.QuantizedDnn.loadFromFile(p2)
    O, O2 = dnn.outputDimension(), dnn2.outputDimension()
    x = torch.from_numpy(F.synth_features(n, 432, seed=41)).cuda()
    ref = torch.empty((n, O), dtype=torch.float32, device="cuda")
    ref2 = torch.empty((n, O2), dtype=torch.float32, device="cuda")
    dnn.calculate_device(x.data_ptr(), n, ref.data_ptr(), 0)
    dnn2.calculate_device(x.data_ptr(), n, ref2.data_ptr(), 0)
    torch.cuda.synchronize()
    assert abs(float(ref[:64].sum(1).mean()) - 1.0) < 1e-3
    outs = [torch.empty((n, O), dtype=torch.float32, device="cuda") for _ in range(T)]
    out2 = torch.empty((n, O2), dtype=torch.float32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(T + 1)]
    times = [[] for _ in range(T + 1)]
    bad = []
    go = threading.Barrier(T + 1)

    def caller(t):
        s = streams[t]
        go.wait()
        for _ in range(rounds):
            t0 = time.perf_counter()
            if t < T:
                dnn.calculate_device(x.data_ptr(), n, outs[t].data_ptr(), s.cuda_stream)
            else:
                dnn2.calculate_device(x.data_ptr(), n, out2.data_ptr(), s.cuda_stream)
            s.synchronize()
            times[t].append(time.perf_counter() - t0)
            got, want = (outs[t], ref) if t < T else (out2, ref2)
            if not torch.equal(got, want):
                bad.append(t)

    th = [threading.Thread(target=caller, args=(t,)) for t in range(T + 1)]
    for h in th:
        h.start()
    for h in th:
        h.join()
    assert not bad, f"threads {sorted(set(bad))} saw results that differ from the single-stream ones"
    if not os.environ.get("FDNN_FUSE_NORM"):
        assert dnn.fuseGiveups() == 0 and dnn2.fuseGiveups() == 0
    allt = sorted(v for t in range(T) for v in times[t][2:])
    # (nine callers share one GPU: a call takes ~9 single-stream times; no cliff.  The 97th percentile, not the 99th: the 384
    # samples come from nine Python threads, and on a busy host a scheduling hiccup of 15 ms in a handful of them -- once in
    # some 200 runs of this test -- is not the latency cliff this line is about, which would hit most calls; the give-up
    # counter above is the exact detector of that mechanism)
    med, p97 = allt[len(allt) // 2], allt[int(len(allt) * 0.97)]
    # (advisor, round 5: wall-clock bounds flake on loaded boxes.  The correctness gate is the give-up counter above; the
    # latency line fails only on a cliff-sized spread, and reports anything past the old bound without failing)
    if p97 >= 3.0 * med + 2e-3:
        import warnings

        warnings.warn(f"many-streams latency spread: median {med * 1e3:.2f} ms, 97th percentile {p97 * 1e3:.2f} ms, slowest {[round(v * 1e3, 2) for v in allt[-5:]]}")
    assert p97 < 10.0 * med + 20e-3, (med, p97, allt[-5:])
    dnn.delete()
    dnn2.delete()


def test_new_contexts_while_other_callers_keep_the_device_busy(net_model_path):
    """A context's scratch counters (layer 0's list of flagged outputs, the fused soft-max's arrival counters) are zeroed
    when the context is made, and that must have HAPPENED before its first kernel runs on a caller's non-blocking stream
    (make_ctx waits for its null-stream memsets).  Contexts are made on a model's first calls, so: a fresh model, eight
    caller threads starting at once on their own streams, three 10 000-frame calls each, twelve times over -- every result
    equal to the single-stream one, no tile of the fused soft-max given up.  (Before the wait: one failure in ten of the
    many-streams test, a few hundred rows with layer-0 bytes left at their screened value.)"""
    import threading

    import torch

    n, T = 10000, 8
    x = torch.from_numpy(F.synth_features(n, 432, seed=43)).cuda()
    streams = [torch.cuda.Stream() for _ in range(T)]
    ref = None
    for rep in range(12):
        dnn = api.QuantizedDnn.loadFromFile(net_model_path)
        O = dnn.outputDimension()
        if ref is None:
            ref = torch.empty((n, O), dtype=torch.float32, device="cuda")
            dnn.calculate_device(x.data_ptr(), n, ref.data_ptr(), 0)
            torch.cuda.synchronize()
            dnn.delete()
            dnn = api.QuantizedDnn.loadFromFile(net_model_path)  # (an empty context pool again)
        outs = [torch.zeros((n, O), dtype=torch.float32, device="cuda") for _ in range(T)]
        torch.cuda.synchronize()
        bad = []
        go = threading.Barrier(T)

        def caller(t):
            go.wait()
            for r in range(3):
                dnn.calculate_device(x.data_ptr(), n, outs[t].data_ptr(), streams[t].cuda_stream)
                streams[t].synchronize()
                with torch.cuda.stream(streams[t]):
                    if not torch.equal(outs[t], ref):
                        bad.append((t, r))

        th = [threading.Thread(target=caller, args=(t,)) for t in range(T)]
        for h in th:
            h.start()
        for h in th:
            h.join()
        assert not bad, (rep, bad)
        if not os.environ.get("FDNN_FUSE_NORM"):
            assert dnn.fuseGiveups() == 0, rep
        dnn.delete()


@pytest.mark.parametrize("n", [10000, 1000, 100])
def test_deferred_ordering_records_between_streams(net_model_path, n):
    """A context whose work went to a stream that outlives it (its own, the null stream) only notes that and records its
    hand-over event when ANOTHER stream next uses it (fdnn_runtime.cpp: ctx_enter / ctx_leave, the device's chain of fused
    launches in run_output); work on a caller-created stream is recorded at once.  One thread, no synchronisation between
    the calls, the pooled context -- and its scratch buffers -- handed from stream to stream: null stream -> caller
    stream -> host call (the context's own stream) -> null stream -> second caller stream, different frames each time.
    Every result equals the one of the same call made alone.  (CalculationContext is per call in the reference,
    dnn.cc:146-165: nothing to order there.)"""
    import torch

    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    O = dnn.outputDimension()
    xs = [torch.from_numpy(F.synth_features(n, 432, seed=70 + i)).cuda() for i in range(5)]
    refs = []
    for x in xs:  # each call alone
        r = torch.empty((n, O), dtype=torch.float32, device="cuda")
        dnn.calculate_device(x.data_ptr(), n, r.data_ptr(), 0)
        torch.cuda.synchronize()
        refs.append(r)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        outs = [torch.zeros((n, O), dtype=torch.float32, device="cuda") for _ in range(5)]
        torch.cuda.synchronize()
        dnn.calculate_device(xs[0].data_ptr(), n, outs[0].data_ptr(), 0)               # null stream: noted, not recorded
        dnn.calculate_device(xs[1].data_ptr(), n, outs[1].data_ptr(), sa.cuda_stream)  # caller stream: the deferred record, then its own
        host = dnn.calculate(xs[2].cpu().numpy())                                        # the context's own stream
        dnn.calculate_device(xs[3].data_ptr(), n, outs[3].data_ptr(), 0)
        dnn.calculate_device(xs[4].data_ptr(), n, outs[4].data_ptr(), sb.cuda_stream)
        torch.cuda.synchronize()
        for i in (0, 1, 3, 4):
            assert torch.equal(outs[i], refs[i]), (rep, i)
        assert np.array_equal(host, refs[2].cpu().numpy()), rep
    if not os.environ.get("FDNN_FUSE_NORM"):
        assert dnn.fuseGiveups() == 0
    dnn.delete()


@pytest.mark.parametrize("out_dim,n", [(8000, 10000), (1003, 3000), (8000, 100), (1003, 37)])
def test_bit_mask_entry_points_equal_the_byte_mask_ones(net_model_path, tmp_models, out_dim, n):
    """fdnn_ctx_lazy_output_batch_bits[_device]: the LazyContext contract (dnn.cc:355-392) with the active set handed over
    as bits, one 64-bit word per 64 nodes -- no 80 MB of mask bytes per 10 000-frame step, no pack pass.  Large batches
    read the words as they are (fused and unfused masked instances, odd output widths), small ones unpack them for the
    small-batch kernels.  Device and host forms, bit for bit the byte-mask results; a sample against the oracle."""
    import torch

    if out_dim == 8000:
        path = net_model_path
    else:
        path = os.path.join(tmp_models, f"bits_{out_dim}.bin")
        F.write_model_bin(path, F.synth_net([432, 256, 256, 256, out_dim], seed=13))
    x = F.synth_features(n, 432, seed=52)
    masks = F.generate_masks(n, out_dim, 0.40, 0.03, seed=9)
    bits = F.pack_mask_bits(masks)
    assert bits.shape == (n, (out_dim + 63) // 64) and bits.dtype == np.uint64
    assert int(bits[0, 0]) & 0xff == int(np.packbits(masks[0, :8] != 0, bitorder="little")[0])
    dnn = api.QuantizedDnn.loadFromFile(path)
    ctx = dnn.getNewLazyContext(n)
    xd = torch.from_numpy(x).cuda()
    md = torch.from_numpy(masks).cuda()
    bd = torch.from_numpy(bits.view(np.int64)).cuda()
    a = torch.zeros((n, out_dim), dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    s = torch.cuda.current_stream().cuda_stream
    ctx.calculateUntilOutputDevice(xd.data_ptr(), s)
    ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), a.data_ptr(), 0, n, s)
    ctx.calculateForOutputNodesBatchBitsDevice(bd.data_ptr(), b.data_ptr(), 0, n, s)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    k = min(n, 300)   # host form, a range in the middle of the context
    first = (n - k) // 2
    got = ctx.calculateForOutputNodesBatchBits(bits[first:first + k], first)
    assert np.array_equal(got, a[first:first + k].cpu().numpy())
    idx = np.arange(0, n, max(1, n // 64))
    want = Oracle(path).lazy(x[idx], masks[idx])
    assert np.abs(a[torch.from_numpy(idx).cuda()].cpu().numpy() - want).max() <= TIGHT
    ctx.delete()
    dnn.delete()


@pytest.mark.parametrize("out_dim,n,frac", [(8000, 100, 0.4), (8000, 1000, 0.4), (1003, 300, 0.1), (1003, 64, 0.9), (130, 50, 0.5)])
def test_host_lazy_batches_return_compacted_rows(net_model_path, tmp_models, out_dim, n, frac):
    """A host caller of the batched lazy contract gets its rows back compacted: inactive nodes all read 1 / total
    (dnn.cc:366-369, :389), so only the active probabilities and that value cross PCIe and the rows are rebuilt inside the
    caller's array.  Byte and bit forms against LazyOutputActivations (dnn.cc:355-392), with rows that are all active, all
    inactive, and one node short of either; identical to the uncompacted device result bit for bit."""
    import torch

    if out_dim == 8000:
        path = net_model_path
    else:
        path = os.path.join(tmp_models, f"compact_{out_dim}.bin")
        F.write_model_bin(path, F.synth_net([432, 128, 128, 128, out_dim], seed=17))
    x = F.synth_features(n, 432, seed=53)
    masks = F.generate_masks(n, out_dim, frac, 0.05, seed=3)
    masks[0] = 1
    masks[1] = 0
    masks[2] = 1
    masks[2, out_dim - 1] = 0
    masks[3] = 0
    masks[3, 0] = 1
    dnn = api.QuantizedDnn.loadFromFile(path)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    got = ctx.calculateForOutputNodesBatch(masks)
    want = Oracle(path).lazy(x, masks)
    assert np.abs(got - want).max() <= TIGHT
    got_bits = ctx.calculateForOutputNodesBatchBits(F.pack_mask_bits(masks))
    assert np.array_equal(got, got_bits)
    # the device-resident (uncompacted) result of the same call
    md = torch.from_numpy(masks).cuda()
    od = torch.zeros((n, out_dim), dtype=torch.float32, device="cuda")
    ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, n, 0)
    torch.cuda.synchronize()
    assert np.array_equal(od.cpu().numpy(), got)
    # a sub-range of the context
    sub = ctx.calculateForOutputNodesBatch(masks[10:40], 10)
    assert np.array_equal(sub, got[10:40])
    ctx.delete()
    dnn.delete()


@pytest.mark.parametrize("out_dim,n", [(1003, 10000), (3483, 5000), (2001, 12000)])
def test_fused_softmax_with_arbitrary_output_widths(tmp_models, out_dim, n):
    """Round 5: output widths that are not a multiple of 32 (real pdf counts; the reference takes any width, dnn.cc:428-454)
    scale their soft-max inside the output kernel as well (ANYW fused instance) instead of falling back to the scale pass.
    Large batches, dense and lazy with bit masks: bit-identical to the unfused path, rows sum to one, sampled frames equal
    the oracle; and the give-up path of the instance (FDNN_GEMM_DEBUG=4096 is covered for width 8000 above) is exercised
    through the accumulator-probe / properties here."""
    import torch

    p = os.path.join(tmp_models, f"fusew_{out_dim}.bin")
    F.write_model_bin(p, F.synth_net([432, 256, 256, 256, out_dim], seed=out_dim))
    x = F.synth_features(n, 432, seed=out_dim + 1)
    masks = F.generate_masks_fast(n, out_dim, 0.4, 0.03, seed=5)
    bits = F.pack_mask_bits(masks)
    dnn = api.QuantizedDnn.loadFromFile(p)
    xd = torch.from_numpy(x).cuda()
    bd = torch.from_numpy(bits.view(np.int64)).cuda()
    outs = {}
    for mode in (0, 1):
        api.set_fuse(mode)
        od = torch.zeros((n, out_dim), dtype=torch.float32, device="cuda")
        ol = torch.zeros((n, out_dim), dtype=torch.float32, device="cuda")
        dnn.calculate_device(xd.data_ptr(), n, od.data_ptr(), 0)
        dnn.calculate_lazy_bits_device(xd.data_ptr(), n, bd.data_ptr(), ol.data_ptr(), 0)
        torch.cuda.synchronize()
        outs[mode] = (od, ol)
    api.set_fuse(-1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[1][0].sum(1, dtype=torch.float64) - 1).abs().max()) < 1e-4
    assert float((outs[1][1].sum(1, dtype=torch.float64) - 1).abs().max()) < 1e-4
    idx = sample_every_tile(n, 320, 2, seed=3)[:96]
    orc = Oracle(p)
    want = orc.calculate(x[idx])
    got = outs[1][0][torch.from_numpy(idx).cuda()].cpu().numpy()
    assert np.abs(got - want).max() <= TIGHT
    lwant = orc.lazy(x[idx], masks[idx])
    lgot = outs[1][1][torch.from_numpy(idx).cuda()].cpu().numpy()
    assert np.abs(lgot - lwant).max() <= TIGHT
    assert dnn.fuseGiveups() == 0
    dnn.delete()


@pytest.mark.parametrize("n", [10000, 12000])
def test_layers_without_saturating_pairs_take_the_walk_free_instances(tmp_models, n):
    """A layer whose list of saturating pairs is empty (trained, heavy-tailed nets) runs k-loops without the entry walk
    (qgemm_kernel / qchain_kernel NOFIX: hidden layers per launch at 10 000 frames, chained at 12 000; the fused dense and
    masked output instances).  Two nets of production width: no layer with pairs, and pairs in the third layer and the
    output layer only (so that the chained kernel keeps its walk while the per-layer launches mix both kinds).  Last hidden
    layer bit for bit and probabilities (dense and 40 % masks) against the oracle on sampled frames; both hidden-layer
    forms equal."""
    topo = [432, 1024, 1024, 1024, 1024, 2048]
    clean = F.synth_net(topo, seed=31, mode="nosat")
    mixed = F.synth_net(topo, seed=31, mode="nosat")
    gauss = F.synth_net(topo, seed=31, mode="gauss")
    for li in (2, 4):
        mixed.layers[li] = gauss.layers[li]
    x = F.synth_features(n, 432, seed=19)
    masks = F.generate_masks_fast(n, topo[-1], 0.40, 0.03, seed=23)
    idx = np.array(sorted(set(np.linspace(0, n - 1, 24).astype(int)) | {319, 320, n - 1}))
    for name, net in (("clean", clean), ("mixed", mixed)):
        p = os.path.join(tmp_models, f"nofix_{name}.bin")
        F.write_model_bin(p, net)
        pairs = [api.HostModel(p).risky_pairs(j) for j in range(1, len(topo) - 1)]
        assert (sum(pairs) == 0) == (name == "clean"), pairs
        if name == "mixed":
            assert pairs[1] > 0 and pairs[0] == 0
        dnn = api.QuantizedDnn.loadFromFile(p)
        orc = Oracle(p)
        want, taps = orc.calculate(x[idx], taps=True)
        hid = {}
        for mode in (0, 1):
            api.set_chain(mode, 1)
            ctx = dnn.getNewLazyContext(n)
            ctx.calculateUntilOutput(x)
            hid[mode] = ctx.hiddenActivations()[idx].copy()
            lazy = ctx.calculateForOutputNodesBatch(masks)[idx].copy() if mode == 0 else None
            ctx.delete()
            if lazy is not None:
                assert np.abs(lazy - orc.lazy(x[idx], masks[idx])).max() <= TIGHT, name
        api.set_chain(-1)
        assert np.array_equal(hid[0], taps["u8_acts"][-1]) and np.array_equal(hid[1], hid[0]), name
        got = dnn.calculate(x)[idx]
        assert np.abs(got - want).max() <= TIGHT, name
        dnn.delete()
