"""CPU-only checks of the product library: it loads, exports every symbol
include/*.h declares, its load-time half (loader, quantizer, LUT, per-node
offsets) matches the golden vectors / oracle, and it refuses to compute without
a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden
from fast_dnn_amd import api, formats as F
from oracle.oracle import Oracle


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(fdnn_[a-z0-9_]+|Java_suskun_nn_QuantizedDnn_[A-Za-z]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    L = api.lib()
    decl = _declared("fdnn.h") | _declared("fdnn_jni.h")
    assert len(decl) >= 50
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    # and the Python binding covers the whole header
    assert _declared("fdnn.h") == set(api.SIGNATURES), _declared("fdnn.h") ^ set(api.SIGNATURES)
    assert _declared("fdnn_jni.h") == set(api.JNI_SYMBOLS)


def test_sigmoid_lut_matches_reference_bytes():
    assert (api.host_sigmoid_lut() == golden("lut.npz")["lut"]).all()


def test_quantizer_edge_cases_match_reference():
    g = golden("quantizer.npz")
    for name in g["names"]:
        wq, mult = api.host_quantize(g[f"{name}_w"], float(g[f"{name}_cut"]))
        assert (wq == g[f"{name}_wq"]).all(), name
        ref = float(g[f"{name}_mult"])
        assert mult == ref or (np.isinf(mult) and np.isinf(ref)), name


def test_host_model_matches_golden_and_oracle(tiny_model_path):
    g = golden("tiny.npz")
    hm = api.HostModel(tiny_model_path)
    o = Oracle(tiny_model_path)
    assert hm.n_layers == 4
    assert [hm.layer_in(j) for j in range(4)] == [432, 64, 64, 64]
    assert [hm.layer_out(j) for j in range(4)] == [64, 64, 64, 100]
    wq = np.concatenate([hm.weights_q(j).ravel() for j in range(1, 4)])
    assert (wq == g["wq"]).all()
    assert [hm.multiplier(j) for j in range(1, 4)] == list(g["mult"])
    for j in range(1, 4):
        assert (hm.bias(j) == o.layer_bias(j)).all()
        assert (hm.wsum128(j) == 128 * hm.weights_q(j).astype(np.int64).sum(1)).all()
        assert hm.risky_pairs(j) == o.risky_pairs(j)
    assert hm.blob_size() % 256 == 0


def test_unpadded_input_dim_is_padded_like_the_reference(tmp_path):
    p = str(tmp_path / "m429.bin")
    F.write_model_bin(p, F.synth_net([429, 64, 64, 64, 100], seed=2))
    hm = api.HostModel(p)
    assert hm.layer_in(0) == 432  # float_dnn.cc:32-33
    assert Oracle(p).in_dim == 432


def test_full_net_quantization_hashes(net_model_path):
    import hashlib

    g = golden("net_full.npz")
    hm = api.HostModel(net_model_path)
    assert [hm.multiplier(j) for j in range(1, hm.n_layers)] == list(g["mult"])
    sh = [hashlib.sha256(hm.weights_q(j).tobytes()).hexdigest() for j in range(1, hm.n_layers)]
    assert sh == list(g["wq_sha256"])
    assert [hm.risky_pairs(j) for j in range(1, hm.n_layers)] == list(g["risky_pairs"])


@pytest.mark.parametrize("topo,why", [
    ([432, 64, 64, 100], "4 affine"),           # dnn.cc:199: layers()[1] must be hidden
    ([432, 64, 48, 64, 100], "same width"),
    ([432, 72, 72, 72, 100], "multiple of 16"),
])
def test_rejects_nets_the_reference_cannot_run(tmp_path, topo, why):
    p = str(tmp_path / "bad.bin")
    F.write_model_bin(p, F.synth_net(topo, seed=1))
    with pytest.raises(api.FdnnError) as e:
        api.HostModel(p)
    assert e.value.code == api.FDNN_E_FORMAT


def test_error_paths(tmp_path, tiny_model_path):
    with pytest.raises(api.FdnnError) as e:
        api.HostModel(str(tmp_path / "missing.bin"))
    assert e.value.code == api.FDNN_E_IO
    trunc = str(tmp_path / "trunc.bin")
    open(trunc, "wb").write(open(tiny_model_path, "rb").read()[:5000])
    with pytest.raises(api.FdnnError) as e:
        api.HostModel(trunc)
    assert e.value.code == api.FDNN_E_FORMAT
    with pytest.raises(ValueError):
        api.QuantizedDnn.loadFromFile(tiny_model_path, weightCutOffValue=0.0)


def test_very_large_batches_are_cut_into_whole_rounds():
    """Chunking of very large passes (the device-side counterpart of the reference's frame blocks, dnn.cc:402-454:
    blocking never changes a result): the chunks tile [0, n) in order, none exceeds 20 480 frames, every chunk but the last
    is 20 480 frames, and up to 20 480 frames a pass is one batch (round 5: the chained hidden layers took the partial
    rounds of workgroups away, so a tail is no longer split off)."""
    for n in list(range(1, 200, 7)) + [10239, 10240, 10241, 11000, 12288, 12289, 15000, 20480, 20481, 22528, 22529, 25000, 31000,
                                        33000, 125000, 1000000, 20971520]:
        ch = api.frame_chunks(n)
        assert ch[0][0] == 0 and sum(c for _, c in ch) == n
        assert all(a[0] + a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all(0 < c <= 20480 for _, c in ch) and ch[0][1] == max(c for _, c in ch)
        assert all(c == 20480 for _, c in ch[:-1])
    assert api.frame_chunks(10240) == [(0, 10240)]
    assert api.frame_chunks(11000) == [(0, 11000)]
    assert api.frame_chunks(15000) == [(0, 15000)]
    assert api.frame_chunks(33000) == [(0, 20480), (20480, 12520)]
    assert api.frame_chunks(31000) == [(0, 20480), (20480, 10520)]
    # hidden layers launched layer by layer (chaining off / not possible): up to 2 048 frames past a whole round go alone
    assert api.frame_chunks(11000, chained=False) == [(0, 10240), (10240, 760)]
    assert api.frame_chunks(10240, chained=False) == [(0, 10240)]
    assert api.frame_chunks(9000, chained=False) == [(0, 9000)]
    assert api.frame_chunks(15000, chained=False) == [(0, 15000)]  # 4 760 past the round: worth its own round
    assert api.frame_chunks(31000, chained=False) == [(0, 20480), (20480, 10240), (30720, 280)]


def test_no_cpu_fallback_without_a_gpu(tiny_model_path):
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.FdnnError) as e:
        api.QuantizedDnn.loadFromFile(tiny_model_path)
    assert e.value.code == api.FDNN_E_DEVICE and "no CPU path" in str(e.value)
    with pytest.raises(api.FdnnError) as e:  # the group entry points sit on the same loader
        api.DeviceGroup(tiny_model_path, [0, 0])
    assert e.value.code == api.FDNN_E_DEVICE


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under fast-dnn_amd/ may reference it."""
    pkg = os.path.join(ROOT, "fast-dnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, os.path.join(dirpath, f)


def test_received_blob_is_bounds_checked(tiny_model_path):
    """adopt_blob (what a rank does with the broadcast weights): every section offset, the padded
    dimensions and the saturation fix-up lists are validated before they can drive device reads."""
    import struct

    blob = api.HostModel(tiny_model_path).blob()
    assert api.host_blob_check(blob)["hidden_dim"] == 64
    hdr_q0 = 96  # BlobHeader: 16 bytes magic/version/size, 8 ints/floats, 6 u64 offsets, then QLayerDesc q[]

    def rejects(mutate):
        b = blob.copy()
        mutate(b)
        with pytest.raises(api.FdnnError) as e:
            api.host_blob_check(b)
        assert e.value.code == api.FDNN_E_FORMAT

    rejects(lambda b: b.__setitem__(slice(0, 4), np.frombuffer(b"XXXX", np.uint8)))           # magic
    with pytest.raises(api.FdnnError):
        api.host_blob_check(blob[:-256])                                                         # truncated
    off_w0 = 16 + 8 * 4  # magic, version, total_bytes, 8 ints/floats
    assert struct.unpack_from("<Q", blob, off_w0)[0] % 256 == 0 and struct.unpack_from("<Q", blob, off_w0)[0] < blob.size
    rejects(lambda b: struct.pack_into("<Q", b, off_w0, blob.size - 256))                        # W0 runs past the end
    # first int8 layer descriptor: 5 u64 offsets, then rows, rows_pad, cols, n_fix
    assert struct.unpack_from("<iii", blob, hdr_q0 + 40) == (64, 256, 64)
    rejects(lambda b: struct.pack_into("<Q", b, hdr_q0, blob.size - 512))                        # weight rows past the end
    rejects(lambda b: struct.pack_into("<i", b, hdr_q0 + 44, 200))                               # rows_pad not a tile multiple
    rejects(lambda b: struct.pack_into("<i", b, hdr_q0 + 52, 1 << 28))                           # n_fix beyond the entry list


def test_layer_with_millions_of_saturating_pairs_loads(tmp_path):
    """The reference runs any weights; a layer whose every pair can saturate (all weights equal ->
    all 127 after quantization) used to be refused at 2^20 listed pairs."""
    net = F.synth_net([432, 2048, 2048, 2048, 64], seed=3)
    net.layers[1].weights[:] = 0.5
    p = str(tmp_path / "allsat.bin")
    F.write_model_bin(p, net)
    hm = api.HostModel(p)
    assert hm.risky_pairs(1) == 2048 * 1024 > (1 << 20)
    assert hm.risky_pairs(1) == Oracle(p).risky_pairs(1)
    assert api.host_blob_check(hm.blob())["n_affine"] == 4


@pytest.mark.parametrize("O", [8000, 1000, 77, 64])
def test_compacted_lazy_rows_expand_to_the_reference_layout(O):
    """The host half of a compacted lazy return: every inactive node of a row reads the row's one inactive value
    (exp(0) / total, dnn.cc:366-369, :389), the active ones their own -- the library's expanding-load form, its scalar
    form and the in-place form (compacted rows in the tail of the caller's block) against plain numpy indexing."""
    import ctypes as C

    rng = np.random.default_rng(O)
    n = 37
    masks = (rng.random((n, O)) < 0.4).astype(np.int8)
    masks[0] = 0
    masks[1] = 1
    masks[2, : O // 2] = 1
    masks[3, -1] = 1
    masks[4, -1] = 0
    bits = F.pack_mask_bits(masks)
    most = int(masks.sum(1).max())
    for stride in (most + 1, min(O + 1, most + 9)):
        comp = rng.random((n, stride), dtype=np.float32)
        want = np.empty((n, O), dtype=np.float32)
        for f in range(n):
            want[f] = comp[f, 0]
            want[f, masks[f] != 0] = comp[f, 1 : 1 + int(masks[f].sum())]
        for mode in (0, 1, 2):
            if mode == 2 and stride > O:
                continue
            out = np.full((n, O), -3.0, dtype=np.float32)
            api._check(api.lib().fdnn_debug_lazy_expand(out.ctypes.data_as(api._c_f32p), comp.ctypes.data_as(api._c_f32p), n, O, stride,
                                                        C.c_void_p(bits.ctypes.data), mode))
            assert np.array_equal(out, want), (O, stride, mode)
