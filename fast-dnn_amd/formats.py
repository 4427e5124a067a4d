"""File formats of the fast-dnn scorer and seeded synthetic nets / feature batches.

Nothing here touches the GPU.  Two on-disk formats, both big-endian (they are
written by Java ``DataOutputStream`` in the reference):

* ``.bin`` model  -- reference writer ``FeedForwardNetwork.saveBinary``
  (src/java/suskun/nn/FeedForwardNetwork.java:226-235, layer body :331-340),
  reference reader ``FloatDnn::FloatDnn`` (src/cpp/float_dnn.cc:18-69):

      i32 layerCount
      per layer:  i32 inDim, i32 outDim, outDim*inDim f32 weights (node-major
                  rows), outDim f32 bias
      inDim0 f32 shift, inDim0 f32 scale

* feature matrix -- writer ``BatchData.serializeDataMatrix``
  (src/java/suskun/nn/BatchData.java:107-139), reader ``BatchData(fileName)``
  (src/cpp/float_dnn.cc:85-105):

      i32 frames, i32 dim, frames*dim f32

  The reference CLI writes its *output* matrix host-endian (little) with u32
  header (src/cpp/float_dnn.cc:114-164); ``read_output_matrix`` reads that.

The synthetic generator is seeded (numpy PCG64) so that a 170 MB net is
reproducible on the GPU box from a seed; tests pin a sha256 of a small net.
"""
from __future__ import annotations

import hashlib
import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


@dataclass
class FloatLayerSpec:
    """One affine layer: ``weights[out, in]`` (node-major rows) and ``bias[out]``."""

    weights: np.ndarray
    bias: np.ndarray

    @property
    def in_dim(self) -> int:
        return int(self.weights.shape[1])

    @property
    def out_dim(self) -> int:
        return int(self.weights.shape[0])


@dataclass
class FloatNet:
    """In-memory fp32 net as stored in a ``.bin`` model file."""

    layers: List[FloatLayerSpec]
    shift: np.ndarray
    scale: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def input_dim(self) -> int:
        return self.layers[0].in_dim

    @property
    def output_dim(self) -> int:
        return self.layers[-1].out_dim

    def topology(self) -> List[int]:
        return [self.input_dim] + [l.out_dim for l in self.layers]


# --------------------------------------------------------------------------- model .bin


def write_model_bin(path: str, net: FloatNet) -> None:
    """Write ``net`` in the reference ``.bin`` layout (all big-endian)."""
    with open(path, "wb") as f:
        f.write(struct.pack(">i", len(net.layers)))
        for layer in net.layers:
            w = np.ascontiguousarray(layer.weights, dtype=np.float32)
            b = np.ascontiguousarray(layer.bias, dtype=np.float32)
            assert w.ndim == 2 and b.shape == (w.shape[0],)
            f.write(struct.pack(">ii", w.shape[1], w.shape[0]))
            f.write(w.astype(">f4").tobytes())
            f.write(b.astype(">f4").tobytes())
        f.write(np.asarray(net.shift, dtype=np.float32).astype(">f4").tobytes())
        f.write(np.asarray(net.scale, dtype=np.float32).astype(">f4").tobytes())


def read_model_bin(path: str) -> FloatNet:
    """Read a ``.bin`` model exactly as stored (no padding applied)."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def i32() -> int:
        nonlocal off
        (v,) = struct.unpack_from(">i", buf, off)
        off += 4
        return v

    def f32(n: int) -> np.ndarray:
        nonlocal off
        a = np.frombuffer(buf, dtype=">f4", count=n, offset=off).astype(np.float32)
        off += 4 * n
        return a

    n_layers = i32()
    layers = []
    in0 = None
    for j in range(n_layers):
        in_dim, out_dim = i32(), i32()
        if j == 0:
            in0 = in_dim
        w = f32(in_dim * out_dim).reshape(out_dim, in_dim)
        b = f32(out_dim)
        layers.append(FloatLayerSpec(w, b))
    shift = f32(in0)
    scale = f32(in0)
    if off != len(buf):
        raise ValueError(f"{path}: {len(buf) - off} trailing bytes")
    return FloatNet(layers, shift, scale)


def model_bin_size(topology: Sequence[int]) -> int:
    """Byte size of a ``.bin`` for ``[in, h1, ..., out]``."""
    size = 4
    for i in range(1, len(topology)):
        size += 8 + 4 * topology[i] * topology[i - 1] + 4 * topology[i]
    return size + 8 * topology[0]


# --------------------------------------------------------------------------- feature matrices


def write_feature_bin(path: str, frames: np.ndarray) -> None:
    """Big-endian ``i32 n, i32 dim, n*dim f32`` (what the reference CLI reads)."""
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    assert frames.ndim == 2
    with open(path, "wb") as f:
        f.write(struct.pack(">ii", frames.shape[0], frames.shape[1]))
        f.write(frames.astype(">f4").tobytes())


def read_feature_bin(path: str) -> np.ndarray:
    """Read a big-endian feature matrix; trailing rows beyond the header count
    (the reference writer's off-by-one, BatchData.java:126-138) are ignored,
    as the native reader does."""
    with open(path, "rb") as f:
        buf = f.read()
    n, d = struct.unpack_from(">ii", buf, 0)
    if n < 0 or d < 0 or 8 + 4 * n * d > len(buf):
        raise ValueError(f"{path}: header {n}x{d} does not fit {len(buf)} bytes")
    return np.frombuffer(buf, dtype=">f4", count=n * d, offset=8).astype(np.float32).reshape(n, d)


def read_output_matrix(path: str) -> np.ndarray:
    """Host-endian ``u32 n, u32 dim, n*dim f32`` as written by the CLI in BIN mode
    (src/cpp/float_dnn.cc:140-150)."""
    with open(path, "rb") as f:
        buf = f.read()
    n, d = struct.unpack_from("<II", buf, 0)
    return np.frombuffer(buf, dtype="<f4", count=n * d, offset=8).reshape(n, d).copy()


# --------------------------------------------------------------------------- synthetic nets


def synth_net(
    topology: Sequence[int],
    seed: int = 1,
    mode: str = "gauss",
    w0_std: float = 0.02,
    w_std: float = 0.05,
    bias_std: float = 0.1,
) -> FloatNet:
    """Seeded synthetic net for ``topology = [in, h1, ..., out]``.

    Distribution follows SURVEY.md section 8(d) config 3: layer-0 weights
    N(0, 0.02^2), int8 layers N(0, 0.05^2), biases N(0, 0.1^2), shift
    N(0, 0.1^2), scale U(0.05, 0.07).

    ``mode``:
      * ``"gauss"``  -- plain Gaussian weights; after the per-layer abs-max
        quantizer a few adjacent weight pairs are large enough for the
        reference's ``pmaddubsw`` int16 saturation to fire (dnn.cc:337-340).
      * ``"nosat"``  -- one planted outlier per int8 layer sets the layer
        abs-max so that every other |w_q| <= 64; no pair can saturate
        (heavy-tailed, like trained Kaldi nets).
    """
    if mode not in ("gauss", "nosat"):
        raise ValueError(mode)
    rng = np.random.Generator(np.random.PCG64(seed))
    layers = []
    for i in range(1, len(topology)):
        n_in, n_out = int(topology[i - 1]), int(topology[i])
        std = w0_std if i == 1 else w_std
        w = rng.standard_normal((n_out, n_in), dtype=np.float32) * np.float32(std)
        b = rng.standard_normal(n_out, dtype=np.float32) * np.float32(bias_std)
        if mode == "nosat" and i > 1:
            # clip the body to +-3.2 sigma and plant one weight at 2x that:
            # multiplier = round(127 / (6.4 sigma)), so body |w_q| <= 64.
            lim = np.float32(3.2 * std)
            np.clip(w, -lim, lim, out=w)
            w[n_out // 2, n_in // 2] = np.float32(2.0) * lim
            w[n_out // 2, (n_in // 2) ^ 1] = 0.0  # its pmaddubsw pair partner: keeps the pair below the int16 limit
        layers.append(FloatLayerSpec(w, b))
    n_in0 = int(topology[0])
    shift = rng.standard_normal(n_in0, dtype=np.float32) * np.float32(0.1)
    scale = (rng.random(n_in0, dtype=np.float32) * np.float32(0.02) + np.float32(0.05)).astype(np.float32)
    return FloatNet(layers, shift, scale, meta={"seed": seed, "mode": mode, "topology": list(topology)})


NET_TOPOLOGY = [432] + [2048] * 7 + [8000]  # 7x2048 hidden + 8000 outputs (README.md:64)


def synth_features(n: int, dim: int = 432, seed: int = 7, pad_from: Optional[int] = 429) -> np.ndarray:
    """Seeded feature batch shaped like the shipped Kaldi features: roughly
    mean 1.9 / std 19.9 (data/16khz.bin statistics), with columns
    ``pad_from..dim`` zero (429 = 39 x 11 spliced frames, padded to 432)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((n, dim), dtype=np.float32) * np.float32(19.9) + np.float32(1.9)
    if pad_from is not None and pad_from < dim:
        x[:, pad_from:] = 0.0
    return x


def generate_masks(count: int, dimension: int, ratio: float = 0.40, churn: float = 0.03, seed: int = 11) -> np.ndarray:
    """Seeded restatement of ``FuncTest.generateMasks`` (test/java/suskun/nn/FuncTest.java:121-154):
    frame 0 has ``int(dimension*ratio)`` random ones; every later frame turns
    ``int(dimension*churn)`` zeros on and then the same number of ones off."""
    rng = np.random.Generator(np.random.PCG64(seed))
    active = int(dimension * ratio)
    new_active = int(dimension * churn)
    res = np.zeros((count, dimension), dtype=np.int8)

    def set_random(row: np.ndarray, amount: int, val: int) -> None:
        cnt = 0
        while cnt < amount:
            k = int(rng.integers(0, dimension))
            if row[k] != val:
                row[k] = val
                cnt += 1

    set_random(res[0], active, 1)
    for i in range(1, count):
        res[i] = res[i - 1]
        set_random(res[i], new_active, 1)
        set_random(res[i], new_active, 0)
    return res


def generate_masks_fast(count: int, dimension: int, ratio: float = 0.40, churn: float = 0.03, seed: int = 11) -> np.ndarray:
    """Masks with the statistics of ``generate_masks`` (FuncTest.java:121-154: ``ratio`` of the
    nodes active, ``churn * dimension`` nodes switched on and as many switched off per frame) from a
    per-node two-state Markov chain -- P(off -> on) = churn / (1 - ratio), P(on -> off) =
    churn / ratio, stationary share ``ratio`` -- which vectorises over the nodes.  The active count
    fluctuates around ``ratio * dimension`` instead of being pinned to it; for benchmarks, where
    10 000 x 8000 masks must not take 20 s of rejection sampling (callers count the active nodes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p_on, p_off = churn / (1.0 - ratio), churn / ratio
    res = np.empty((count, dimension), dtype=np.int8)
    row = np.zeros(dimension, dtype=bool)
    row[rng.choice(dimension, int(dimension * ratio), replace=False)] = True
    for i in range(count):
        if i:
            u = rng.random(dimension, dtype=np.float32)
            row = np.where(row, u >= p_off, u < p_on)
        res[i] = row
    return res


def sha256_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def ensure_model_file(path: str, topology: Sequence[int], seed: int = 1, mode: str = "gauss") -> str:
    """Write the seeded net to ``path`` unless a file of the right size is already there."""
    want = model_bin_size(topology)
    if not (os.path.exists(path) and os.path.getsize(path) == want):
        tmp = f"{path}.tmp{os.getpid()}"
        write_model_bin(tmp, synth_net(topology, seed=seed, mode=mode))
        os.replace(tmp, path)
    return path


def pack_mask_bits(masks) -> np.ndarray:
    """Byte masks [n][O] (non-zero = active) -> uint64 [n][ceil(O / 64)], bit b of word w = node 64 w + b: the layout of
    fdnn_ctx_lazy_output_batch_bits."""
    m = (np.asarray(masks) != 0)
    n, O = m.shape
    wpr = (O + 63) // 64
    padded = np.zeros((n, wpr * 64), dtype=np.uint8)
    padded[:, :O] = m
    return np.packbits(padded, axis=1, bitorder="little").view("<u8").reshape(n, wpr)
