"""Parity of the HIP path with the CPU oracle and the committed golden vectors.
All calls go through the C-ABI of libfast-dnn.so (ctypes binding in
fast_dnn_amd.api).  Integer state is compared bit-for-bit; soft-max
probabilities to 1e-3 (BASELINE.json north_star), observed ~1e-7."""
import hashlib
import os

import numpy as np
import pytest

from conftest import golden
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star tolerance on soft-max probabilities
TIGHT = 2e-6  # what the implementation actually achieves (expf + sum order)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def x16():
    return golden("tiny.npz")["x16"]


def test_tiny_golden_all_taps(tiny_model_path, x16):
    g = golden("tiny.npz")
    dnn = api.QuantizedDnn.loadFromFile(tiny_model_path)
    assert (dnn.inputDimension(), dnn.outputDimension(), dnn.layerCount()) == (432, 100, 4)
    assert [dnn.layerDimension(k) for k in range(-1, 5)] == [-1, 64, 64, 100, -1, -1]
    t = dnn.forwardTaps(x16)
    assert (t["l0_lin"] == g["l0_lin"]).all()
    assert (t["u8_acts"] == g["u8_acts"]).all()
    assert (t["acc_hid"] == g["acc_hid"]).all()
    assert (t["acc_out"] == g["acc_out"]).all()
    assert np.abs(t["logits"] - g["logits"]).max() <= 1e-6
    assert np.abs(t["probs"] - g["probs"]).max() <= TIGHT
    p = dnn.calculate(x16, 10)
    assert (p == t["probs"]).all()  # taps run the same kernels
    p8 = dnn.calculate(g["x8"], 3)
    assert np.abs(p8 - g["probs8"]).max() <= TIGHT
    dnn.delete()


def test_tiny_fma_flavour(tiny_model_path, x16):
    g = golden("tiny.npz")
    dnn = api.QuantizedDnn.loadFromFile(tiny_model_path)
    dnn.setInputLayerFma(True)
    t = dnn.forwardTaps(x16)
    assert (t["l0_lin"] == g["fma_l0_lin"]).all()
    assert (t["u8_acts"] == g["fma_u8_acts"]).all()
    assert np.abs(t["probs"] - g["fma_probs"]).max() <= TIGHT
    dnn.delete()


def test_fma_flavour_mfma_chains_at_scale(net_model_path):
    """The fused flavour runs on v_mfma_f32_32x32x1_2b_f32: every one of the 4 x 2048 x n
    fmaf chains over 108 k-quads must come out bit-identical to the CPU fmaf chains, for a
    frame count that leaves a ragged 128-frame tile."""
    x = F.synth_features(333, 432, seed=77)
    Oracle.set_l0_fma(True)
    try:
        want, wt = Oracle(net_model_path).calculate(x, taps=True)
    finally:
        Oracle.set_l0_fma(False)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    dnn.setInputLayerFma(True)
    t = dnn.forwardTaps(x)
    assert (t["l0_lin"] == wt["l0_lin"]).all()
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    dnn.setInputLayerFma(False)
    t2 = dnn.forwardTaps(x)
    assert not (t2["l0_lin"] == wt["l0_lin"]).all()  # the canonical unfused flavour differs in the last ulp
    dnn.delete()


def test_saturation_fixture(sat_model_path):
    """pmaddubsw pair saturation fires (dnn.cc:337-340): the sparse correction must
    reproduce the reference's int32 sums exactly."""
    g = golden("sat.npz")
    dnn = api.QuantizedDnn.loadFromFile(sat_model_path)
    t = dnn.forwardTaps(g["x"])
    assert int(g["sat_events"]) > 0
    assert (t["u8_acts"] == g["u8_acts"]).all()
    assert (t["acc_hid"] == g["acc_hid"]).all()
    assert (t["acc_out"] == g["acc_out"]).all()
    assert np.abs(t["probs"] - g["probs"]).max() <= TIGHT
    dnn.delete()


def test_mid_lazy_golden(mid_model_path, x16):
    g = golden("mid_lazy.npz")
    masks = F.generate_masks(100, 1000, 0.40, 0.03, seed=11)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    ctx = dnn.getNewLazyContext(100)  # FuncTest.lazyEmulation protocol (FuncTest.java:92-119)
    ctx.calculateUntilOutput(x16)
    assert (ctx.hiddenActivations() == g["hidden_last"]).all()
    rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(40)])
    assert np.abs(rows - g["lazy"]).max() <= TIGHT
    off = rows[0][masks[0] == 0]
    assert off.min() > 0 and np.allclose(off, off[0])  # masked-out nodes come back as 1/total
    batch = ctx.calculateForOutputNodesBatch(masks[:40])
    assert (batch == rows).all()
    sub = ctx.calculateForOutputNodesBatch(masks[13:29], first=13)  # unaligned sub-range
    assert (sub == rows[13:29]).all()
    ctx.delete()
    dense = dnn.calculate(x16)
    assert np.abs(dense[:20] - g["dense"]).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("frames", [8, 16, 32])
def test_decoder_sized_lazy_blocks_on_the_full_net(net_model_path, frames):
    """The batched alternative to the per-frame calculateLazy (one JNI round trip and ~35 us per frame through the host
    API, README.md:45): a decoder hands over a block of 8..32 frames with their masks (FuncTest.generateMasks: 40 %
    active, 3 % churn per frame).  Full 8000-output net, small-batch kernels, against the oracle's
    LazyOutputActivations (dnn.cc:355-392) -- and bit for bit against the per-frame entry point."""
    n = frames
    x = F.synth_features(n, 432, seed=123 + frames)
    masks = F.generate_masks(n, 8000, 0.40, 0.03, seed=17)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    block = ctx.calculateForOutputNodesBatch(masks)
    rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(n)])
    ctx.delete()
    dnn.delete()
    want = Oracle(net_model_path).lazy(x, masks)
    assert np.abs(block - want).max() <= TIGHT
    assert np.array_equal(block, rows)
    off = block[0][masks[0] == 0]
    assert off.min() > 0 and np.all(off == off[0])  # masked-out nodes: exp(0) / total each


def test_full_net_hashes(net_model_path):
    g = golden("net_full.npz")
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    assert dnn.layerCount() == 8 and dnn.outputDimension() == 8000
    t = dnn.forwardTaps(g["x"])
    assert [sha(t["u8_acts"][j]) for j in range(7)] == list(g["u8_sha256"])
    assert [sha(t["acc_hid"][j]) for j in range(6)] == list(g["acc_hid_sha256"])
    assert sha(t["acc_out"]) == str(g["acc_out_sha256"])
    assert np.abs(t["probs"][:4] - g["probs4"]).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("n", [1, 2, 63, 127, 128, 129, 300, 1000])
def test_ragged_batch_sizes_vs_oracle(mid_model_path, n):
    x = F.synth_features(n, 432, seed=100 + n)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    want, wt = Oracle(mid_model_path).calculate(x, taps=True)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_hid"] == wt["acc_hid"]).all()
    assert (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    assert np.abs(dnn.calculate(x) - want).max() <= TIGHT
    dnn.delete()


def test_empty_input_and_argument_errors(tiny_model_path):
    dnn = api.QuantizedDnn.loadFromFile(tiny_model_path)
    assert dnn.calculate(np.zeros((0, 432), np.float32)).shape == (0, 0)  # QuantizedDnn.java:154-156
    with pytest.raises(ValueError):
        dnn.calculate(np.zeros((3, 429), np.float32))  # QuantizedDnn.java:157-161
    ctx = dnn.getNewLazyContext(4)
    with pytest.raises(api.FdnnError) as e:
        ctx.calculateForOutputNodes(np.ones(100, np.int8))  # before calculateUntilOutput
    assert e.value.code == api.FDNN_E_STATE
    ctx.calculateUntilOutput(F.synth_features(4))
    for _ in range(4):
        ctx.calculateForOutputNodes(np.ones(100, np.int8))
    with pytest.raises(api.FdnnError) as e:
        ctx.calculateForOutputNodes(np.ones(100, np.int8))  # frame index past the end
    assert e.value.code == api.FDNN_E_ARG
    ctx.delete()
    dnn.delete()


def test_input_is_not_modified(tiny_model_path, x16):
    """The reference shifts/scales the caller's buffer in place (dnn.cc:175-192); this path must not."""
    dnn = api.QuantizedDnn.loadFromFile(tiny_model_path)
    x = x16.copy()
    dnn.calculate(x)
    assert (x == x16).all()
    dnn.delete()


def test_nosat_net_has_no_fixups_and_matches(tmp_models):
    import os

    p = os.path.join(tmp_models, "nosat.bin")
    F.write_model_bin(p, F.synth_net([432, 256, 256, 256, 500], seed=9, mode="nosat"))
    hm = api.HostModel(p)
    assert [hm.risky_pairs(j) for j in range(1, 4)] == [0, 0, 0]
    x = F.synth_features(200, seed=5)
    dnn = api.QuantizedDnn.loadFromFile(p)
    want, wt = Oracle(p).calculate(x, taps=True)
    t = dnn.forwardTaps(x)
    assert (t["acc_hid"] == wt["acc_hid"]).all() and (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    dnn.delete()


def test_masked_dense_equivalences(mid_model_path):
    """Size-independent properties: all-ones mask == dense; all-zero mask == uniform 1/O."""
    x = F.synth_features(130, seed=77)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    ctx = dnn.getNewLazyContext(130)
    ctx.calculateUntilOutput(x)
    ones = ctx.calculateForOutputNodesBatch(np.ones((130, 1000), np.int8))
    assert (ones == dnn.calculate(x)).all()
    zeros = ctx.calculateForOutputNodesBatch(np.zeros((130, 1000), np.int8))
    assert np.allclose(zeros, 1.0 / 1000, rtol=1e-6)
    ctx.delete()
    dnn.delete()


def test_full_size_properties(net_model_path):
    """BASELINE config-3 shape (7x2048 -> 8000, 10k frames): soft-max rows sum to 1,
    results are independent of how the batch is split (frames are independent), and a
    sample of rows matches the oracle."""
    x = F.synth_features(10000, seed=21)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    p = dnn.calculate(x)
    assert p.shape == (10000, 8000)
    assert np.abs(p.sum(1, dtype=np.float64) - 1).max() < 1e-4
    assert (p >= 0).all()
    head = dnn.calculate(x[:777])
    assert (head == p[:777]).all()
    tail = dnn.calculate(x[9000:])
    assert (tail == p[9000:]).all()
    idx = np.array([0, 1, 4999, 9998, 9999])
    want = Oracle(net_model_path).calculate(x[idx])
    assert np.abs(p[idx] - want).max() <= TIGHT
    dnn.delete()


def test_concurrent_callers_share_one_model(mid_model_path):
    """MultiThreadedStressTest.java:48-69: many threads, one QuantizedDnn."""
    from concurrent.futures import ThreadPoolExecutor

    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    orc = Oracle(mid_model_path)
    xs = [F.synth_features(20 + 17 * i, seed=300 + i) for i in range(12)]
    want = [orc.calculate(x) for x in xs]

    def task(i):
        return dnn.calculate(xs[i % len(xs)])

    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(task, range(96)))
    for i, gp in enumerate(got):
        assert np.abs(gp - want[i % len(xs)]).max() <= TIGHT
    dnn.delete()


def test_true_divide_variant_on_degenerate_layer(tmp_models):
    """An all-zero int8 layer gives multiplier = round(127/0) = inf (dnn.cc:479): the
    3-op division cannot be validated, the layer must run the IEEE-divide kernel
    variant and still match the oracle bit for bit (0/inf = 0, then + bias)."""
    import os

    net = F.synth_net([432, 128, 128, 128, 200], seed=12)
    net.layers[2].weights[:] = 0.0  # second int8 hidden layer
    p = os.path.join(tmp_models, "zero_layer.bin")
    F.write_model_bin(p, net)
    x = F.synth_features(150, seed=6)
    orc = Oracle(p)
    assert np.isinf(orc.layer_mult(2))
    want, wt = orc.calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_hid"] == wt["acc_hid"]).all() and (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    dnn.delete()


def test_huge_bias_takes_the_exact_table_path(tmp_models):
    """|lin|*200 may exceed 2^31 -> the half-step table shortcut is not provably valid;
    the layer must fall back to the exact round()/INT_MIN semantics of dnn.h:36-43."""
    import os

    net = F.synth_net([432, 128, 128, 128, 200], seed=13)
    net.layers[1].bias[::7] = 3.0e7   # (int)round(100*x) overflows int32 -> x86 gives INT_MIN -> entry 0
    net.layers[1].bias[3::7] = -3.0e7
    p = os.path.join(tmp_models, "huge_bias.bin")
    F.write_model_bin(p, net)
    x = F.synth_features(70, seed=8)
    want, wt = Oracle(p).calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("hidden", [64, 144, 272])
def test_hidden_widths_not_multiple_of_the_k_step(tmp_models, hidden):
    """Hidden widths are only required to be x16 (README.md:10): the k padding to 128 and
    the 256-node tile padding must be invisible."""
    import os

    p = os.path.join(tmp_models, f"h{hidden}.bin")
    F.write_model_bin(p, F.synth_net([432, hidden, hidden, hidden, 52], seed=20 + hidden))
    x = F.synth_features(333, seed=9)
    want, wt = Oracle(p).calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_hid"] == wt["acc_hid"]).all() and (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("in_dim,frames", [(20, 129), (64, 77), (100, 300), (440, 256), (448, 130)])
def test_input_widths_and_the_chain_image_padding(tmp_models, in_dim, frames):
    """Layer 0 runs on chain-major images padded to whole 12- or 16-row chunks and 128-column
    tiles (fdnn_l0.hip): widths that pick either chunk depth, with and without a zero tail, and
    frame counts around the 128-frame tile must give the reference's layer-0 sums bit for bit."""
    import os

    p = os.path.join(tmp_models, f"in{in_dim}.bin")
    F.write_model_bin(p, F.synth_net([in_dim, 144, 144, 144, 40], seed=60 + in_dim))
    x = F.synth_features(frames, in_dim, seed=in_dim)
    want, wt = Oracle(p).calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    for kind in (1, 2, 0):  # chain-pass kernel, 64 x 64-tile kernel, the library's own choice
        dnn.setInputLayerKernel(kind)
        t = dnn.forwardTaps(x)
        assert np.array_equal(t["l0_lin"].view(np.uint32), wt["l0_lin"].view(np.uint32)), kind
        assert (t["u8_acts"] == wt["u8_acts"]).all(), kind
        assert np.abs(t["probs"] - want).max() <= TIGHT
        # production kernels (no taps) agree with the tapped instance
        assert np.array_equal(dnn.calculate(x), t["probs"]), kind
    dnn.delete()


def test_cli_matches_oracle(tmp_path, mid_model_path, x16):
    """fast-dnn model input out BIN|TXT (dnn.cc:20-84): big-endian input matrix, host-endian
    u32-header BIN output, one text row per frame."""
    import subprocess

    cli = os.path.join(os.path.dirname(api.LIB_PATH), "fast-dnn")
    inp = str(tmp_path / "in.bin")
    F.write_feature_bin(inp, x16)
    out_bin, out_txt = str(tmp_path / "o.bin"), str(tmp_path / "o.txt")
    r = subprocess.run([cli, mid_model_path, inp, out_bin, "BIN"], capture_output=True, text=True, check=True)
    assert "Network = 432-2x256-1000" in r.stdout and "Input   = 100x432" in r.stdout  # PrintTopology undercounts by one
    subprocess.run([cli, mid_model_path, inp, out_txt, "TXT"], check=True, capture_output=True)
    want = Oracle(mid_model_path).calculate(x16, batch=8)
    got = F.read_output_matrix(out_bin)
    assert got.shape == (100, 1000) and np.abs(got - want).max() <= TIGHT
    txt = np.loadtxt(out_txt, dtype=np.float32)
    assert txt.shape == (100, 1000) and np.abs(txt - want).max() <= 1e-5  # 6 significant digits
    assert subprocess.run([cli, mid_model_path], capture_output=True).returncode != 0
    # ... and against the reference's own CLI (oracle/_ref/fast-dnn, compiled from the reference's sources where they
    # were available: it travels with the repo as a binary): same console lines, same output file layout, same numbers
    ref_cli = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "fast-dnn")
    if os.path.exists(ref_cli):
        ref_bin, ref_txt = str(tmp_path / "r.bin"), str(tmp_path / "r.txt")
        rr = subprocess.run([ref_cli, mid_model_path, inp, ref_bin, "BIN"], capture_output=True, text=True, check=True)
        subprocess.run([ref_cli, mid_model_path, inp, ref_txt, "TXT"], check=True, capture_output=True)
        keep = lambda text: [ln for ln in text.splitlines() if ln.startswith(("Network", "Input   "))]
        assert keep(r.stdout) == keep(rr.stdout), (r.stdout, rr.stdout)
        ref = F.read_output_matrix(ref_bin)
        assert ref.shape == got.shape and np.abs(got - ref).max() <= TIGHT
        with open(out_bin, "rb") as fa, open(ref_bin, "rb") as fb:
            assert fa.read(8) == fb.read(8)  # host-endian u32 frame / dimension header (float_dnn.cc:140-150)
        ref_t = np.loadtxt(ref_txt, dtype=np.float32)
        assert ref_t.shape == txt.shape and np.abs(txt - ref_t).max() <= 1e-5


def test_config1_16khz_frames_tiled_to_1000_on_the_full_net(net_model_path, x16):
    """BASELINE configs[1], literally: the shipped data/16khz.bin frames (100 valid rows; golden x16) tiled x10 to a
    1000-frame batch, full soft-max, on the full 432 -> 7x2048 -> 8000 net, against the oracle:
    last hidden layer bit for bit, probabilities within 2e-6; and every repetition of a frame gets the same bits."""
    x = np.tile(x16, (10, 1))
    assert x.shape == (1000, 432)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    got = dnn.calculate(x)
    ctx = dnn.getNewLazyContext(1000)
    ctx.calculateUntilOutput(x)
    hid = ctx.hiddenActivations()
    ctx.delete()
    dnn.delete()
    want, wt = Oracle(net_model_path).calculate(x16, taps=True)   # the 100 distinct frames
    assert np.array_equal(hid[:100], wt["u8_acts"][-1])
    assert np.abs(got[:100] - want).max() <= TIGHT
    for r in range(1, 10):
        assert np.array_equal(got[100 * r:100 * r + 100], got[:100]) and np.array_equal(hid[100 * r:100 * r + 100], hid[:100])


def test_quantization_error_against_the_float_net(mid_model_path, x16):
    """SURVEY 8(f) row 4 -- the reference's accuracy notion (FuncTest.diff, FuncTest.java:59-74):
    distance of the quantized scorer to the unquantized fp32 net (FeedForwardNetwork.calculate)."""
    from fast_dnn_amd import convert as CV

    net = F.read_model_bin(mid_model_path)
    ref = CV.float_forward(net, x16)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    q = dnn.calculate(x16)
    dnn.delete()
    rep = CV.quantization_report(ref, q)
    want = CV.quantization_report(ref, Oracle(mid_model_path).calculate(x16))
    assert rep["nodes_over_0.1"] == want["nodes_over_0.1"]          # same verdict as the CPU reference algorithm
    assert abs(rep["mean_abs_diff"] - want["mean_abs_diff"]) < 1e-6
    assert rep["top1_agreement"] > 0.8 and rep["max_abs_diff"] < 0.2, rep


def test_shard_of_the_million_frame_config(net_model_path):
    """BASELINE configs[4]: 1 M frames over 8 GPUs = 125 000 frames per device in one call
    (4 GB of probabilities).  Spot frames must equal the oracle, every row must sum to one, and
    the batch must agree with the same frames scored in a small batch (no dependence on n)."""
    import torch

    n = 125_000
    base = F.synth_features(1000, 432, seed=31)
    xd = torch.from_numpy(base).cuda().repeat(n // 1000, 1).contiguous()
    out = torch.empty((n, 8000), dtype=torch.float32, device="cuda")
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    dnn.calculate_device(xd.data_ptr(), n, out.data_ptr(), 0)
    torch.cuda.synchronize()
    sums = out.sum(1)
    assert float((sums - 1).abs().max()) < 1e-4
    small = dnn.calculate(base[:8])
    for rep in (0, 57, 124):  # the 1000-frame block repeats: rows rep*1000 + i equal rows i
        assert np.abs(out[rep * 1000: rep * 1000 + 8].cpu().numpy() - small).max() == 0
    want = Oracle(net_model_path).calculate(base[:2])
    assert np.abs(out[124_000:124_002].cpu().numpy() - want).max() <= TIGHT
    dnn.delete()


def test_shard_of_the_million_frame_config_one_row_in_eight(net_model_path):
    """configs[4], one full 125 000-frame shard of DISTINCT frames: the last hidden layer's u8 activations of every eighth row
    (15 625 rows: every 320-frame tile, every position modulo 8 inside the 32-frame MFMA blocks over the shard) against the
    oracle bit for bit, and the probabilities of every 64th row to 2e-6 (dnn.cc:402-454)."""
    n = 125_000
    x = F.synth_features(n, 432, seed=131)
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    hid = ctx.hiddenActivations()
    orc = Oracle(net_model_path)
    rows = np.arange(0, n, 8) + (np.arange(0, n, 8) // 8) % 8  # stride 8 with a rotating phase: all positions mod 8
    rows = rows[rows < n]
    want = orc.hidden_acts_mt(x[rows])
    assert np.array_equal(hid[rows], want)
    sub = rows[::8]
    ctx.delete()
    p = dnn.calculate(x[sub])  # (a batch of its own: scoring does not depend on the batch a frame arrives in -- asserted above)
    assert np.abs(p - orc.output_mt(want[::8])).max() <= TIGHT
    dnn.delete()


@pytest.mark.parametrize("hidden", [64, 144, 400])
def test_big_batch_small_net_takes_the_8_wave_shapes(tmp_models, hidden):
    """Enough frames that the 256/320-frame, 8-wave kernel shapes (rotated-barrier k-loop) are
    chosen, with k-loops of only 1, 2 and 4 steps: prologue/refill edge cases of that loop."""
    import os

    p = os.path.join(tmp_models, f"wide_batch_h{hidden}.bin")
    F.write_model_bin(p, F.synth_net([432, hidden, hidden, hidden, 300], seed=40 + hidden))
    n = 40000  # one 256-node tile needs > 32 768 frames before the 128-frame shape stops being chosen
    x = F.synth_features(n, 432, seed=12)
    want, wt = Oracle(p).calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    got = dnn.calculate(x)
    assert np.abs(got - want).max() <= TIGHT
    t = dnn.forwardTaps(x[:700])  # taps for a slice (the tap kernels are separate instances)
    assert (t["u8_acts"] == wt["u8_acts"][:, :700]).all()
    assert (t["acc_out"] == wt["acc_out"][:700]).all()
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    assert (ctx.hiddenActivations() == wt["u8_acts"][-1]).all()   # last hidden layer, all 40 000 frames, bit for bit
    ctx.delete()
    dnn.delete()


def test_one_frame_lazy_call_equals_the_batched_path(net_model_path, sat_model_path):
    """The per-frame JNI call (one 32-frame small-batch tile, 31 rows of padding) must give bit-for-bit
    what the batch gives for the same frame, and the oracle's LazyOutputActivations numbers -- on the
    full 2048 -> 8000 layer and on the net whose pairs really saturate."""
    x = F.synth_features(24, 432, seed=41)
    masks = F.generate_masks(24, 8000, 0.40, 0.03, seed=5)
    masks[3] = 0   # nothing active: every node comes back as 1/8000
    masks[4] = 1   # everything active
    dnn = api.QuantizedDnn.loadFromFile(net_model_path)
    ctx = dnn.getNewLazyContext(24)
    ctx.calculateUntilOutput(x)
    batch = ctx.calculateForOutputNodesBatch(masks)            # 24 > 8 frames: device staging instead of pinned
    rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(24)])
    assert (rows == batch).all()
    assert np.allclose(rows[3], 1.0 / 8000, rtol=0, atol=1e-9)
    want = Oracle(net_model_path).lazy(x[:6], masks[:6])
    assert np.abs(rows[:6] - want).max() <= TIGHT
    ctx.delete()
    dnn.delete()

    g = golden("sat.npz")
    dnn = api.QuantizedDnn.loadFromFile(sat_model_path)
    xs = g["x"]
    n, O = xs.shape[0], dnn.outputDimension()
    ones = np.ones((n, O), np.int8)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(xs)
    rows = np.stack([ctx.calculateForOutputNodes(ones[i]) for i in range(n)])
    assert np.abs(rows - g["probs"]).max() <= TIGHT            # all-active lazy == dense, saturation included
    assert (rows == dnn.calculate(xs)).all() or np.abs(rows - dnn.calculate(xs)).max() <= 1e-9
    ctx.delete()
    dnn.delete()


@pytest.mark.parametrize("out_dim", [1000, 1001, 1030])
def test_small_lazy_blocks_equal_the_larger_batch(tmp_models, sat_model_path, out_dim):
    """Decoder-sized lazy blocks (1..16 frames: the host-mapped pinned protocol up to 8, device staging above) against
    the same frames scored inside a 40-frame block: bit for bit the same -- output widths on and off the 16-byte grid,
    ragged frame counts, empty and full masks, any non-zero mask byte, and the net whose pairs really saturate."""
    import os

    p = os.path.join(tmp_models, f"rowwise{out_dim}.bin")
    F.write_model_bin(p, F.synth_net([432, 128, 128, 128, out_dim], seed=300 + out_dim))
    x = F.synth_features(40, seed=out_dim + 7)
    masks = F.generate_masks(40, out_dim, 0.3, 0.05, seed=out_dim + 3)
    masks[1] = 0
    masks[2] = 1
    masks[3] = np.where(masks[3] != 0, -7, 0)  # any non-zero byte is "active"
    dnn = api.QuantizedDnn.loadFromFile(p)
    big = dnn.getNewLazyContext(40)
    big.calculateUntilOutput(x)
    gemm = big.calculateForOutputNodesBatch(masks)
    big.delete()
    assert np.abs(gemm - Oracle(p).lazy(x, masks)).max() <= TIGHT
    for n in (1, 5, 13, 16):
        ctx = dnn.getNewLazyContext(n)
        ctx.calculateUntilOutput(x[:n])
        assert np.array_equal(ctx.calculateForOutputNodesBatch(masks[:n]), gemm[:n]), n
        ctx.delete()
    dnn.delete()

    g = golden("sat.npz")
    dnn = api.QuantizedDnn.loadFromFile(sat_model_path)
    xs = g["x"][:12]
    ones = np.ones((xs.shape[0], dnn.outputDimension()), np.int8)
    ctx = dnn.getNewLazyContext(xs.shape[0])
    ctx.calculateUntilOutput(xs)
    block = ctx.calculateForOutputNodesBatch(ones)
    ctx.delete()
    assert np.abs(block - g["probs"][:12]).max() <= TIGHT
    assert np.array_equal(block, dnn.calculate(xs))  # all-active lazy == dense, saturation included
    dnn.delete()


def test_device_pointer_lazy_api(mid_model_path, x16):
    """fdnn_ctx_forward_hidden_device / fdnn_ctx_lazy_output_batch_device: same numbers as the
    host-pointer calls, buffers owned by the caller (torch only provides the device memory)."""
    import torch

    masks = F.generate_masks(100, 1000, 0.40, 0.03, seed=11)
    dnn = api.QuantizedDnn.loadFromFile(mid_model_path)
    ctx = dnn.getNewLazyContext(100)
    ctx.calculateUntilOutput(x16)
    want = ctx.calculateForOutputNodesBatch(masks)
    ctx.delete()
    s = torch.cuda.Stream()
    xd = torch.from_numpy(x16).cuda()
    md = torch.from_numpy(masks).cuda()
    od = torch.zeros((100, 1000), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx = dnn.getNewLazyContext(100)
    ctx.calculateUntilOutputDevice(xd.data_ptr(), s.cuda_stream)
    ctx.calculateForOutputNodesBatchDevice(md.data_ptr(), od.data_ptr(), 0, 100, s.cuda_stream)
    ctx.calculateForOutputNodesBatchDevice(md[40:].data_ptr(), od[40:].data_ptr(), 40, 17, s.cuda_stream)  # sub-range again
    s.synchronize()
    assert (od.cpu().numpy() == want).all()
    dense = torch.zeros((100, 1000), dtype=torch.float32, device="cuda")
    dnn.calculate_device(xd.data_ptr(), 100, dense.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.abs(dense.cpu().numpy() - dnn.calculate(x16)).max() == 0
    ctx.delete()
    dnn.delete()


def test_blob_export_import_round_trip(mid_model_path, x16):
    """What ranks 1..N-1 of the multi-GPU bench do (fast-dnn_amd/dist.py:load_replicated): a model
    rebuilt from the exported device blob -- including the layer-0 weight image built at import --
    scores bit-identically to the one loaded from the file, in both layer-0 flavours and kernels."""
    import torch

    a = api.QuantizedDnn.loadFromFile(mid_model_path)
    nbytes = a.blobSize()
    blob = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    a.exportBlob(blob.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    b = api.QuantizedDnn.fromDeviceBlob(blob.data_ptr(), nbytes, 0)
    del blob  # the import copies
    x = F.synth_features(300, seed=123)
    for fma in (False, True):
        for kind in (0, 1, 2):
            for d in (a, b):
                d.setInputLayerFma(fma)
                d.setInputLayerKernel(kind)
            assert np.array_equal(a.calculate(x), b.calculate(x)), (fma, kind)
    assert (b.inputDimension(), b.outputDimension(), b.layerCount()) == (a.inputDimension(), a.outputDimension(), a.layerCount())
    b.delete()
    a.delete()


@pytest.mark.parametrize("out_dim", [37, 101, 1003])
def test_output_widths_not_multiple_of_four(tmp_models, out_dim):
    """The output layer is the one layer the reference never pads (FeedForwardNetwork.extend): real
    pdf counts are arbitrary.  Such widths take the general epilogue (scalar mask reads, scalar
    stores) instead of the branch-free dense / masked instances -- dense and lazy, batch and
    single frame, against the oracle."""
    import os

    p = os.path.join(tmp_models, f"out{out_dim}.bin")
    F.write_model_bin(p, F.synth_net([432, 128, 128, 128, out_dim], seed=80 + out_dim))
    x = F.synth_features(77, seed=out_dim)
    masks = F.generate_masks(77, out_dim, 0.4, 0.05, seed=out_dim + 1)
    orc = Oracle(p)
    want, wt = orc.calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    t = dnn.forwardTaps(x)
    assert (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    assert np.array_equal(dnn.calculate(x), t["probs"])
    ctx = dnn.getNewLazyContext(77)
    ctx.calculateUntilOutput(x)
    lazy_want = orc.lazy(x, masks)
    got = ctx.calculateForOutputNodesBatch(masks)
    assert np.abs(got - lazy_want).max() <= TIGHT
    ctx.currentVectorIndex = 5  # the per-frame call walks the context frame by frame (LazyContext.java)
    assert np.array_equal(ctx.calculateForOutputNodes(masks[5]), got[5])
    ctx.delete()
    dnn.delete()


def test_kaldi_text_model_through_the_hip_path(tmp_path):
    """SURVEY 8(f) row 1 end to end: Kaldi nnet1 text + feature_transform (with a <Splice> block)
    -> convert.load_kaldi_nnet_text -> align(4, 16) -> .bin (FuncTest.java:20-28, the reference's
    own preparation recipe) -> scored by the HIP path == the oracle on the same file."""
    from fast_dnn_amd import convert as CV
    from test_convert import _kaldi_text, _transform_text

    net = F.synth_net([39, 21, 21, 21, 13], seed=5)  # unaligned on purpose: 39 -> 40 inputs, 21 -> 32 hidden nodes
    (tmp_path / "final.nnet.txt").write_text(_kaldi_text(net), encoding="utf-8")
    (tmp_path / "final.feature_transform.txt").write_text(_transform_text(net.shift, net.scale, True), encoding="utf-8")
    got = CV.load_kaldi_nnet_text(str(tmp_path / "final.nnet.txt"), str(tmp_path / "final.feature_transform.txt"))
    p = str(tmp_path / "model.bin")
    F.write_model_bin(p, CV.align(got, 4, 16))
    x = CV.align_features(F.synth_features(150, 39, seed=3, pad_from=None), 4)
    want, wt = Oracle(p).calculate(x, taps=True)
    dnn = api.QuantizedDnn.loadFromFile(p)
    assert (dnn.inputDimension(), dnn.hiddenDimension(), dnn.outputDimension()) == (40, 32, 13)
    t = dnn.forwardTaps(x)
    assert (t["u8_acts"] == wt["u8_acts"]).all()
    assert (t["acc_hid"] == wt["acc_hid"]).all() and (t["acc_out"] == wt["acc_out"]).all()
    assert np.abs(t["probs"] - want).max() <= TIGHT
    # the padded hidden nodes have zero weights and zero bias: sigmoid(0) -> table entry 128, fed to zero columns
    assert (t["u8_acts"][:, :, 21:] == 128).all()
    ref = CV.float_forward(CV.align(got, 4, 16), x)
    assert CV.quantization_report(ref, dnn.calculate(x))["max_abs_diff"] < 0.1
    dnn.delete()


def test_one_frame_lazy_call_on_a_net_the_small_kernels_cannot_take(tmp_models):
    """The per-frame JNI call on a net whose last hidden layer is wider than the small-batch GEMM kernel's 2048 inputs: the
    one-frame block then goes through the large-tile output GEMM behind a mask_pack pass that reads the host-mapped
    pinned staging (fdnn_ctx_lazy_output_batch, count <= 8).  Against LazyOutputActivations (dnn.cc:355-392), and
    bit for bit what the 24-frame batch gives for the same frames."""
    import os

    p = os.path.join(tmp_models, "wide_hidden.bin")
    F.write_model_bin(p, F.synth_net([40, 2304, 2304, 2304, 300], seed=29, w_std=0.02))
    n = 24
    x = F.synth_features(n, 40, seed=30, pad_from=None)
    masks = F.generate_masks(n, 300, 0.40, 0.05, seed=6)
    masks[1] = 0
    masks[2] = 1
    dnn = api.QuantizedDnn.loadFromFile(p)
    ctx = dnn.getNewLazyContext(n)
    ctx.calculateUntilOutput(x)
    batch = ctx.calculateForOutputNodesBatch(masks)
    rows = np.stack([ctx.calculateForOutputNodes(masks[i]) for i in range(n)])
    assert (rows == batch).all()
    want = Oracle(p).lazy(x, masks)
    assert np.abs(rows - want).max() <= TIGHT
    assert np.abs(dnn.calculate(x) - Oracle(p).calculate(x)).max() <= TIGHT
    ctx.delete()
    dnn.delete()
