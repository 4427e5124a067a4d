"""Kaldi nnet1 text model / feature text  ->  the scorer's ``.bin`` files  (SURVEY 8(f) row 1).

Python counterpart of the reference's Java-side tooling, which is the only way its users
produce the files the native half reads:

* ``load_kaldi_nnet_text``     <- ``FeedForwardNetwork.loadFromTextFile`` / ``loadLayersFromTextFile``
                                  (src/java/suskun/nn/FeedForwardNetwork.java:86-119, :159-207)
* ``align`` / ``extend``       <- ``FeedForwardNetwork.align`` / ``extend`` (:50-66) and
                                  ``Layer.align`` / ``Layer.extend`` (:262-302)
* ``load_feature_text``        <- ``BatchData.loadMultipleFromText`` (src/java/suskun/nn/BatchData.java:141-180)
* ``align_features`` / ``feature_matrix_bytes`` <- ``BatchData.alignDimension`` (:92-97) and
                                  ``serializeDataMatrix`` (:100-139)

The ``.bin`` writers themselves live in ``formats.py``.  No JDK exists in this image, so the
Java code cannot be run here: the feature-text path is pinned byte-for-byte against the
reference's own ``data/16khz`` -> ``data/16khz.bin`` and ``data/8khz`` -> ``data/8khz.aligned.bin``
pairs (tests/test_convert.py).  No Kaldi text model ships with the reference, so the nnet-text
parser, ``align`` and ``extend`` are pinned by a hand-derived fixture instead: a toy text net whose
expected ``.bin`` bytes were laid out by hand from the Java source, independent of this module
(tests/golden/make_kaldi_toy.py -> tests/golden/kaldi_toy/).

Reference quirks kept on purpose (they shape the files real users have):

* ``serializeDataMatrix(file, n)`` writes ``n + 1`` rows behind a header that says ``n`` when the
  batch holds more than ``n`` rows (``if (k == featureAmount) break; k++`` after the write).
* ``extend`` treats the output layer with ``align(hidden, outputCount)`` -- zero padding up to a
  multiple of ``outputCount`` -- while hidden layers are extended circularly (:58-64).
"""
from __future__ import annotations

import re
import struct
from typing import List, Tuple

import numpy as np

from .formats import FloatLayerSpec, FloatNet

_BLOCK = re.compile(r"\[(.+?)\]", re.DOTALL)
_ID = re.compile(r"(.+?)(?:\[.+?\])", re.DOTALL)


def _parse_f32(tokens) -> np.ndarray:
    """``Float.parseFloat`` of every token: the decimal correctly rounded to float32.  Going through
    a double (strtod, then a cast) rounds twice; the two differ only when the double lands exactly
    on a float32 rounding boundary (low 29 mantissa bits = 1000...0), so just those tokens -- about
    one in 2^29 -- are settled with exact rational arithmetic."""
    d = np.array([float(t) for t in tokens], dtype=np.float64)
    f = d.astype(np.float32)
    if d.size:
        bits = d.view(np.uint64)
        tie = np.flatnonzero(((bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)) & np.isfinite(d) & (np.abs(d) >= 1.2e-38))
        if tie.size:
            from fractions import Fraction

            for i in tie:
                exact = Fraction(tokens[i])
                lo = np.nextafter(np.float32(d[i]), np.float32(-np.inf)) if np.float32(d[i]) > d[i] else np.float32(d[i])
                cands = [lo, np.nextafter(lo, np.float32(np.inf))]
                err = [abs(Fraction(float(c)) - exact) for c in cands]
                if err[0] != err[1]:
                    f[i] = cands[0] if err[0] < err[1] else cands[1]
                else:  # a true tie in the decimal itself: to even
                    f[i] = cands[0] if (cands[0].view(np.uint32) & 1) == 0 else cands[1]
    return f


def _floats(text: str) -> np.ndarray:
    return _parse_f32(text.split())


# ----------------------------------------------------------------------------- model text
def load_kaldi_layers_text(path: str) -> List[FloatLayerSpec]:
    """``<AffineTransform> out in`` headers; every other ``<...>`` line, ``[`` and ``]`` are skipped;
    the first data line starts ``out`` weight rows (brackets stripped) followed by ONE bias row."""
    layers: List[FloatLayerSpec] = []
    with open(path, "r", encoding="utf-8") as fh:
        lines = iter(fh.read().splitlines())
    node_count = in_count = -1
    for raw in lines:
        line = raw.strip()
        if not line:
            continue
        if line.startswith("<AffineTransform>"):
            dims = line[line.index(">") + 1:].split()
            node_count, in_count = int(dims[0]), int(dims[1])
        if node_count == -1 or line.startswith("<") or line in ("[", "]"):
            continue
        rows = []
        cur = line
        for i in range(node_count + 1):
            if i > 0:
                try:
                    cur = next(lines)
                except StopIteration:
                    raise ValueError(f"{path}: layer {len(layers)} ends after {i} of {node_count + 1} rows") from None
            vals = cur.replace("[", " ").replace("]", " ").split()
            want = in_count if i < node_count else node_count
            if len(vals) < want:
                raise ValueError(f"{path}: layer {len(layers)} row {i} has {len(vals)} values, expected {want}")
            rows.append(_parse_f32(vals[:want]))
        layers.append(FloatLayerSpec(np.stack(rows[:node_count]), rows[node_count]))
    if not layers:
        raise ValueError(f"{path}: no <AffineTransform> layer found")
    return layers


def load_feature_transform_text(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Two ``[ ... ]`` blocks = shift (<AddShift>) and scale (<Rescale>); a leading third block
    (<Splice>) is dropped (FeedForwardNetwork.java:97-100)."""
    with open(path, "r", encoding="utf-8") as fh:
        whole = " ".join(fh.read().splitlines())
    blocks = [m.group(1).strip() for m in _BLOCK.finditer(whole)]
    if len(blocks) == 3:
        blocks = blocks[1:]
    if len(blocks) != 2:
        raise ValueError(f"Unexpected feature transformation vector size : {len(blocks)}")
    return _floats(blocks[0]), _floats(blocks[1])


def load_kaldi_nnet_text(network_path: str, transform_path: str) -> FloatNet:
    layers = load_kaldi_layers_text(network_path)
    shift, scale = load_feature_transform_text(transform_path)
    d = layers[0].in_dim
    if shift.size != d:
        raise ValueError(f"Shift transformation vector size {shift.size} is not same as input dimension {d}")
    if scale.size != d:
        raise ValueError(f"Scale transformation vector size {scale.size} is not same as input dimension {d}")
    return FloatNet(layers, shift, scale)


def aligned_size(size: int, alignment: int) -> int:
    return size if size % alignment == 0 else size + alignment - size % alignment


def _pad_vec(v: np.ndarray, alignment: int) -> np.ndarray:
    out = np.zeros(aligned_size(v.size, alignment), dtype=np.float32)
    out[: v.size] = v
    return out


def _align_layer(l: FloatLayerSpec, in_alignment: int, out_alignment: int) -> FloatLayerSpec:
    w = np.zeros((aligned_size(l.out_dim, out_alignment), aligned_size(l.in_dim, in_alignment)), dtype=np.float32)
    w[: l.out_dim, : l.in_dim] = l.weights
    return FloatLayerSpec(w, _pad_vec(l.bias, out_alignment))


def align(net: FloatNet, input_alignment: int = 4, hidden_alignment: int = 16) -> FloatNet:
    """Zero padding: input width to x4 (the SSE float lanes), hidden widths to x16 (one
    ``pmaddubsw`` register of bytes); the output width is left alone."""
    n = len(net.layers)
    layers = []
    for i, l in enumerate(net.layers):
        if i == 0:
            layers.append(_align_layer(l, input_alignment, hidden_alignment if n > 1 else 1))
        elif i < n - 1:
            layers.append(_align_layer(l, hidden_alignment, hidden_alignment))
        else:
            layers.append(_align_layer(l, hidden_alignment, 1))
    return FloatNet(layers, _pad_vec(net.shift, input_alignment), _pad_vec(net.scale, input_alignment))


def _extend_vec(v: np.ndarray, size: int) -> np.ndarray:
    return v[np.arange(size) % v.size].astype(np.float32)


def _extend_layer(l: FloatLayerSpec, in_count: int, out_count: int) -> FloatLayerSpec:
    if out_count < l.out_dim:  # Layer.extend indexes newWeights[i] for every existing node: Java throws here
        raise ValueError(f"extend cannot shrink a layer from {l.out_dim} to {out_count} nodes")
    rows = l.weights[:, np.arange(in_count) % l.in_dim]          # every existing row, circular in k
    w = rows[np.arange(out_count) % l.out_dim].astype(np.float32)  # new nodes copy node i % out_dim
    return FloatLayerSpec(np.ascontiguousarray(w), _extend_vec(l.bias, out_count))


def extend(net: FloatNet, hidden_count: int, output_count: int) -> FloatNet:
    """The author's way of making a big benchmark net out of a small real one: hidden layers grow
    by circular copies; the output layer is only zero-ALIGNED to (hidden_count, output_count)."""
    n = len(net.layers)
    layers = []
    for i, l in enumerate(net.layers):
        if i == 0:
            layers.append(_extend_layer(l, l.in_dim, hidden_count))
        elif i < n - 1:
            layers.append(_extend_layer(l, hidden_count, hidden_count))
        else:
            layers.append(_align_layer(l, hidden_count, output_count))
    return FloatNet(layers, net.shift.copy(), net.scale.copy())


# ----------------------------------------------------------------------------- feature text
def load_feature_text(path: str) -> List[Tuple[str, np.ndarray]]:
    """Kaldi text archive ``utt-id [ row \\n row ... ]``: one (id, frames[n][dim]) per utterance."""
    with open(path, "r", encoding="utf-8") as fh:
        whole = "\n".join(fh.read().splitlines())
    blocks = [m.group(1).strip() for m in _BLOCK.finditer(whole)]
    ids = [m.group(1).strip() for m in _ID.finditer(whole)]
    out = []
    for uid, block in zip(ids, blocks):
        rows = [np.array([float(t) for t in ln.split(" ") if t != ""], dtype=np.float32) for ln in re.split(r"\r|\n", block)]
        out.append((uid, np.stack(rows)))
    return out


def align_features(frames: np.ndarray, alignment: int = 4) -> np.ndarray:
    out = np.zeros((frames.shape[0], aligned_size(frames.shape[1], alignment)), dtype=np.float32)
    out[:, : frames.shape[1]] = frames
    return out


def feature_matrix_bytes(frames: np.ndarray, feature_amount: int = -1, big_endian: bool = True) -> bytes:
    """``BatchData.serializeDataMatrix`` including its off-by-one: the header says
    ``min(feature_amount, n)`` but ``min(feature_amount + 1, n)`` rows follow."""
    n, dim = frames.shape
    if n == 0:
        raise ValueError("There is no data to serialize.")
    if feature_amount < 0:
        feature_amount = n
    feature_amount = min(feature_amount, n)
    rows = min(feature_amount + 1, n)
    order = ">" if big_endian else "<"
    return struct.pack(order + "ii", feature_amount, dim) + np.ascontiguousarray(frames[:rows], dtype=order + "f4").tobytes()


# ----------------------------------------------------------------------------- fp32 reference net
def float_forward(net: FloatNet, frames: np.ndarray) -> np.ndarray:
    """``FeedForwardNetwork.calculate`` (:133-148, sigmoid/softMax :386-414): the UNQUANTIZED
    net -- the reference's only notion of accuracy is the distance to this (FuncTest.diff).
    fp32 accumulation (numpy's blocked order, not Java's sequential loop: ulp-level differences),
    sigmoid and exp evaluated in double and rounded to float as the Java code does."""
    x = ((frames.astype(np.float32) + net.shift) * net.scale).astype(np.float32)
    for i, l in enumerate(net.layers):
        z = (x @ l.weights.T).astype(np.float32) + l.bias
        if i < len(net.layers) - 1:
            x = (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(np.float32)
        else:
            e = np.exp(z.astype(np.float64)).astype(np.float32)
            x = e / e.sum(axis=1, dtype=np.float32, keepdims=True)
    return x


def quantization_report(reference: np.ndarray, quantized: np.ndarray) -> dict:
    """``FuncTest.diff`` (test/java/suskun/nn/FuncTest.java:59-74): per output node, the absolute
    difference summed over the frames; the harness prints every node above 0.1."""
    dif = np.abs(quantized.astype(np.float64) - reference.astype(np.float64)).sum(axis=0)
    return {
        "frames": int(reference.shape[0]),
        "nodes": int(reference.shape[1]),
        "nodes_over_0.1": int((dif > 0.1).sum()),
        "max_node_sum_abs_diff": float(dif.max()),
        "mean_abs_diff": float(np.abs(quantized - reference).mean()),
        "max_abs_diff": float(np.abs(quantized - reference).max()),
        "top1_agreement": float((quantized.argmax(1) == reference.argmax(1)).mean()),
    }
