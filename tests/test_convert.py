"""Kaldi text -> .bin tooling (fast_dnn_amd.convert).  CPU only.

The feature-text path is pinned against the reference's own data pairs (text in, .bin out);
the nnet-text parser can only be checked for self-consistency (no Kaldi text model ships with
the reference and no JDK exists here)."""
import os
import struct

import numpy as np
import pytest

from conftest import golden
from fast_dnn_amd import convert as CV
from fast_dnn_amd import formats as F

REF_DATA = "/root/reference/data"


def test_feature_text_rows_match_the_reference_bin(tmp_path):
    g = golden("feat_text_head.npz")
    for name in ("16khz", "8khz"):
        p = tmp_path / f"{name}.txt"
        p.write_text(str(g[f"{name}_text_head"]), encoding="utf-8")
        (uid, frames), = CV.load_feature_text(str(p))
        assert uid == "1" or uid  # an utterance id precedes the block
        al = CV.align_features(frames, 4)
        want = g[f"{name}_bin_rows"]
        assert al.shape == want.shape and al.dtype == np.float32
        assert (al.view(np.uint32) == want.view(np.uint32)).all()  # bit-for-bit what the Java tooling wrote
        assert (al[:, frames.shape[1]:] == 0).all()


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data files not present")
@pytest.mark.parametrize("text,binary", [("16khz", "16khz.bin"), ("8khz", "8khz.aligned.bin")])
def test_whole_feature_file_byte_identical(text, binary):
    raw = open(os.path.join(REF_DATA, binary), "rb").read()
    n_header, dim = struct.unpack(">ii", raw[:8])
    (_, frames), = CV.load_feature_text(os.path.join(REF_DATA, text))
    got = CV.feature_matrix_bytes(CV.align_features(frames, 4), n_header)
    assert got == raw
    # the off-by-one of serializeDataMatrix shows in the reference's own file whenever it was cut
    rows_in_file = (len(raw) - 8) // (4 * dim)
    assert rows_in_file in (n_header, n_header + 1)
    assert F.read_feature_bin(os.path.join(REF_DATA, binary)).shape == (n_header, dim)


def test_header_vs_rows_quirk():
    x = np.arange(40, dtype=np.float32).reshape(10, 4)
    b = CV.feature_matrix_bytes(x, 3)
    assert struct.unpack(">ii", b[:8]) == (3, 4) and len(b) == 8 + 4 * 4 * 4  # header 3, four rows follow
    assert len(CV.feature_matrix_bytes(x, -1)) == 8 + 10 * 16
    assert len(CV.feature_matrix_bytes(x, 50)) == 8 + 10 * 16
    le = CV.feature_matrix_bytes(x, 2, big_endian=False)
    assert struct.unpack("<ii", le[:8]) == (2, 4)
    with pytest.raises(ValueError):
        CV.feature_matrix_bytes(np.zeros((0, 4), np.float32))


def _kaldi_text(net: F.FloatNet) -> str:
    """What `nnet-copy --binary=false` prints for an nnet1 feed-forward net."""
    out = ["<Nnet> "]
    for i, l in enumerate(net.layers):
        out.append(f"<AffineTransform> {l.out_dim} {l.in_dim} ")
        out.append("<LearnRateCoef> 1 <BiasLearnRateCoef> 1 <MaxNorm> 0  [")
        for r in range(l.out_dim):
            row = "  " + " ".join(repr(float(v)) for v in l.weights[r])
            out.append(row + (" ]" if r == l.out_dim - 1 else " "))
        out.append(" [ " + " ".join(repr(float(v)) for v in l.bias) + " ]")
        out.append(f"<Sigmoid> {l.out_dim} {l.out_dim} " if i < len(net.layers) - 1 else f"<Softmax> {l.out_dim} {l.out_dim} ")
    out.append("</Nnet> ")
    return "\n".join(out) + "\n"


def _transform_text(shift, scale, splice=True) -> str:
    s = "<Nnet> \n"
    if splice:
        s += "<Splice> 429 39 \n[ -5 -4 -3 -2 -1 0 1 2 3 4 5 ]\n"
    s += f"<AddShift> {shift.size} {shift.size} \n<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in shift) + " ]\n"
    s += f"<Rescale> {scale.size} {scale.size} \n<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in scale) + " ]\n</Nnet> \n"
    return s


def test_kaldi_nnet_text_round_trip(tmp_path):
    net = F.synth_net([39, 21, 21, 21, 13], seed=5)  # unaligned on purpose
    (tmp_path / "final.nnet.txt").write_text(_kaldi_text(net), encoding="utf-8")
    for splice in (True, False):
        (tmp_path / "final.feature_transform.txt").write_text(_transform_text(net.shift, net.scale, splice), encoding="utf-8")
        got = CV.load_kaldi_nnet_text(str(tmp_path / "final.nnet.txt"), str(tmp_path / "final.feature_transform.txt"))
        assert len(got.layers) == 4
        for a, b in zip(got.layers, net.layers):
            assert (a.weights == b.weights).all() and (a.bias == b.bias).all()
        assert (got.shift == net.shift).all() and (got.scale == net.scale).all()
    # the reference's own preparation recipe (FuncTest.java:20-28): align(4, 16), saveBinary
    al = CV.align(got, 4, 16)
    assert [l.weights.shape for l in al.layers] == [(32, 40), (32, 32), (32, 32), (13, 32)]
    assert al.shift.size == 40 and al.shift[39] == 0 and al.scale[39] == 0
    assert (al.layers[0].weights[:21, :39] == net.layers[0].weights).all()
    assert (al.layers[0].weights[21:] == 0).all() and (al.layers[0].weights[:, 39:] == 0).all()
    assert (al.layers[3].bias == net.layers[3].bias).all()  # output width untouched
    p = str(tmp_path / "model.bin")
    F.write_model_bin(p, al)
    back = F.read_model_bin(p)
    assert (back.layers[1].weights == al.layers[1].weights).all() and os.path.getsize(p) == F.model_bin_size([40, 32, 32, 32, 13])


def test_transform_errors(tmp_path):
    net = F.synth_net([8, 16, 16, 16, 4], seed=6)
    (tmp_path / "n.txt").write_text(_kaldi_text(net), encoding="utf-8")
    (tmp_path / "t.txt").write_text("<AddShift> 8 8 [ 1 2 3 ]\n", encoding="utf-8")
    with pytest.raises(ValueError, match="Unexpected feature transformation vector size : 1"):
        CV.load_kaldi_nnet_text(str(tmp_path / "n.txt"), str(tmp_path / "t.txt"))
    (tmp_path / "t.txt").write_text(_transform_text(net.shift[:5], net.scale, splice=False), encoding="utf-8")
    with pytest.raises(ValueError, match="Shift transformation vector size 5 is not same as input dimension 8"):
        CV.load_kaldi_nnet_text(str(tmp_path / "n.txt"), str(tmp_path / "t.txt"))
    (tmp_path / "empty.txt").write_text("<Nnet>\n</Nnet>\n", encoding="utf-8")
    with pytest.raises(ValueError, match="no <AffineTransform>"):
        CV.load_kaldi_layers_text(str(tmp_path / "empty.txt"))


def test_extend_is_circular_for_hidden_and_zero_aligned_for_output():
    net = F.synth_net([8, 16, 16, 16, 6], seed=7)
    big = CV.extend(net, 40, 10)
    assert [l.weights.shape for l in big.layers] == [(40, 8), (40, 40), (40, 40), (10, 40)]
    w1 = net.layers[1].weights
    assert (big.layers[1].weights[:16, :16] == w1).all()
    assert (big.layers[1].weights[:16, 16:32] == w1).all() and (big.layers[1].weights[:16, 32:] == w1[:, :8]).all()
    assert (big.layers[1].weights[16:32] == big.layers[1].weights[:16]).all()   # node i copies node i % 16
    assert (big.layers[1].bias[16:32] == net.layers[1].bias).all()
    # the output layer is only ALIGNED (FeedForwardNetwork.java:64): zeros, not copies
    assert (big.layers[3].weights[:6, :16] == net.layers[3].weights).all()
    assert (big.layers[3].weights[6:] == 0).all() and (big.layers[3].weights[:, 16:] == 0).all()
    assert (big.layers[3].bias[6:] == 0).all()


def test_float_reference_and_quantization_report(tiny_model_path):
    """FuncTest's accuracy notion: quantized output vs the fp32 net, per-node abs diff summed over
    frames (nodes above 0.1 are the ones the reference harness prints)."""
    from oracle.oracle import Oracle

    g = golden("tiny.npz")
    net = F.read_model_bin(tiny_model_path)
    ref = CV.float_forward(net, g["x16"])
    assert ref.shape == (100, 100) and np.allclose(ref.sum(1), 1.0, atol=1e-5)
    q = Oracle(tiny_model_path).calculate(g["x16"])
    rep = CV.quantization_report(ref, q)
    assert rep["frames"] == 100 and rep["nodes"] == 100
    assert rep["max_abs_diff"] < 0.05 and rep["top1_agreement"] > 0.9, rep


TOY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_toy")


def test_kaldi_text_to_bin_matches_the_hand_derived_bytes(tmp_path):
    """f1 pin (no JDK here): tests/golden/make_kaldi_toy.py lays the expected files out by hand from
    FeedForwardNetwork.java:50-66, :86-119, :159-207, :226-235, :262-302 -- <Splice> block dropped,
    align(4, 16), extend(5, 4) with its zero-ALIGNED output layer, big-endian saveBinary -- without
    touching fast_dnn_amd.convert.  The converter must reproduce them byte for byte."""
    net = CV.load_kaldi_nnet_text(os.path.join(TOY, "final.nnet.txt"), os.path.join(TOY, "final.feature_transform.txt"))
    assert [l.weights.shape for l in net.layers] == [(2, 3), (2, 2), (3, 2)]
    p = str(tmp_path / "aligned.bin")
    F.write_model_bin(p, CV.align(net, 4, 16))
    assert open(p, "rb").read() == open(os.path.join(TOY, "aligned_4_16.bin"), "rb").read()
    p = str(tmp_path / "extended.bin")
    F.write_model_bin(p, CV.extend(net, 5, 4))
    assert open(p, "rb").read() == open(os.path.join(TOY, "extended_5_4.bin"), "rb").read()
    with pytest.raises(ValueError):
        CV.extend(net, 1, 4)  # Layer.extend cannot shrink (ArrayIndexOutOfBounds in the reference)
    # and the fixture really is what the generator writes (it is committed, not regenerated in CI)
    import subprocess
    import sys

    gen = os.path.join(os.path.dirname(TOY), "make_kaldi_toy.py")
    before = {f: open(os.path.join(TOY, f), "rb").read() for f in os.listdir(TOY)}
    subprocess.check_call([sys.executable, gen], stdout=subprocess.DEVNULL)
    assert before == {f: open(os.path.join(TOY, f), "rb").read() for f in os.listdir(TOY)}


def test_decimal_tokens_round_once_like_float_parsefloat():
    """Float.parseFloat rounds the decimal to float32 ONCE; strtod-then-cast rounds twice and is
    wrong when the double lands exactly on a float32 boundary."""
    # 1 + 2^-24 + 2^-60: just above the midpoint of 1.0 and 1.0 + 2^-23 -> must round UP; as a double it IS the midpoint (-> even = 1.0)
    tok = "1.00000005960464477626025342456"
    assert float(tok) == 1.0 + 2.0 ** -24 and np.float32(float(tok)) == np.float32(1.0)
    got = CV._parse_f32([tok, "0.1", "-7.5e-2", "1.000000059604644775390625"])  # the last one IS the exact tie -> even
    assert got[0] == np.nextafter(np.float32(1.0), np.float32(2.0))
    assert got[1].view(np.uint32) == 0x3DCCCCCD and got[2].view(np.uint32) == 0xBD99999A
    assert got[3] == np.float32(1.0)
