"""File formats (.bin model, feature matrices) and the seeded generators."""
import os
import struct

import numpy as np
import pytest

from conftest import golden
from fast_dnn_amd import formats as F


def test_model_bin_roundtrip(tmp_path):
    net = F.synth_net([429, 64, 64, 64, 100], seed=2)  # unpadded input is legal in the file
    p = str(tmp_path / "m.bin")
    F.write_model_bin(p, net)
    assert os.path.getsize(p) == F.model_bin_size([429, 64, 64, 64, 100])
    back = F.read_model_bin(p)
    assert back.topology() == [429, 64, 64, 64, 100]
    for a, b in zip(net.layers, back.layers):
        assert (a.weights == b.weights).all() and (a.bias == b.bias).all()
    assert (net.shift == back.shift).all() and (net.scale == back.scale).all()
    raw = open(p, "rb").read()
    assert struct.unpack(">iii", raw[:12]) == (4, 429, 64)  # big-endian header


def test_tiny_fixture_is_the_seeded_net(tiny_model_path):
    net = F.synth_net([432, 64, 64, 64, 100], seed=3)
    back = F.read_model_bin(tiny_model_path)
    assert (net.layers[2].weights == back.layers[2].weights).all()


def test_feature_bin_roundtrip_and_trailing_rows(tmp_path):
    x = F.synth_features(7, 432)
    assert (x[:, 429:] == 0).all()
    p = str(tmp_path / "f.bin")
    F.write_feature_bin(p, x)
    assert (F.read_feature_bin(p) == x).all()
    # the reference writer's off-by-one leaves an extra physical row (BatchData.java:126-138)
    with open(p, "ab") as f:
        f.write(np.zeros(432, dtype=">f4").tobytes())
    assert F.read_feature_bin(p).shape == (7, 432)
    with open(p, "r+b") as f:
        f.write(struct.pack(">i", 1000))
    with pytest.raises(ValueError):
        F.read_feature_bin(p)


def test_masks_protocol():
    m = F.generate_masks(20, 8000, 0.40, 0.03, seed=11)
    assert m.dtype == np.int8 and m.shape == (20, 8000)
    assert (m.sum(1) == 3200).all()  # 240 on, then 240 off per frame
    flips = (m[1:] != m[:-1]).sum(1)
    assert (flips <= 480).all() and (flips > 400).all()


def test_nosat_mode_has_no_risky_pairs():
    net = F.synth_net([432, 128, 128, 128, 200], seed=4, mode="nosat")
    for layer in net.layers[1:]:
        w = layer.weights
        mult = np.round(np.float32(127) / np.abs(np.clip(w, -3, 3)).max())
        wq = np.round(w * mult)
        assert np.abs(wq).max() == 127 or np.abs(wq).max() == 128
        body = np.sort(np.abs(wq).ravel())[:-1]
        assert body.max() <= 64
