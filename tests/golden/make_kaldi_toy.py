#!/usr/bin/env python3
"""Hand-derived fixture for the Kaldi-text -> .bin converter.

No JDK exists in this image, so the reference's FeedForwardNetwork cannot be run.  This script
does NOT call fast_dnn_amd.convert either: it lays the expected files out by hand, from literal
tables, following the Java code line by line --

  text parse     FeedForwardNetwork.java:159-207  <AffineTransform> out in; every other '<...>' line,
                 '[' and ']' skipped; `out` weight rows then ONE bias row; brackets stripped
  transform      :86-119   three [...] blocks -> the first (<Splice>) is dropped; shift, scale
  align(4, 16)   :50-58, Layer.align :262-277   first layer (in x4, out x16), middle layers (x16, x16),
                 output layer (in x16, out x1); zero padding; shift / scale padded to x4
  extend(5, 4)   :60-66, Layer.extend :279-302  first layer extend(in, 5), middle layers extend(5, 5):
                 row i >= out copies row i % out, columns k >= in copy column k % in, bias circular;
                 the OUTPUT layer is only align(5, 4)'ed: zero padding to multiples of (5, 4)
  saveBinary     :226-235, Layer.saveToStream :331-340   big-endian i32 layer count; per layer i32 in,
                 i32 out, out rows of in f32, out f32 bias; then shift, scale

Every float is written as the literal IEEE-754 single-precision bit pattern of the decimal token
in the text file (Float.parseFloat = correctly rounded), listed in HEX below; the script asserts
the table against struct for typos only.  Output: tests/golden/kaldi_toy/{final.nnet.txt,
final.feature_transform.txt, aligned_4_16.bin, extended_5_4.bin}.
"""
import os
import struct

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kaldi_toy")

HEX = {  # decimal token -> big-endian float32 bits
    "0": "00000000", "0.1": "3DCCCCCD", "-0.25": "BE800000", "0.5": "3F000000", "3": "40400000", "1e-3": "3A83126F",
    "-7.5e-2": "BD99999A", "2": "40000000", "-1": "BF800000", "0.3333333": "3EAAAAAA", "1.5": "3FC00000",
    "-2.5": "C0200000", "0.75": "3F400000", "4": "40800000", "-0.125": "BE000000", "10": "41200000", "0.2": "3E4CCCCD",
    "-0.3": "BE99999A", "6.25e-2": "3D800000", "0.7": "3F333333", "-1.75": "BFE00000", "0.05": "3D4CCCCD",
    "8": "41000000", "-0.6": "BF19999A", "0.9": "3F666666", "1": "3F800000", "0.007": "3BE56042", "-0.004": "BB83126F",
    "0.0625": "3D800000", "0.015625": "3C800000",
}
for tok, hx in HEX.items():
    assert struct.pack(">f", float(tok)).hex().upper() == hx, tok

W0 = [["0.1", "-0.25", "0.5"], ["3", "1e-3", "-7.5e-2"]]
B0 = ["2", "-1"]
W1 = [["0.3333333", "1.5"], ["-2.5", "0.75"]]
B1 = ["4", "-0.125"]
W2 = [["10", "0.2"], ["-0.3", "6.25e-2"], ["0.7", "-1.75"]]
B2 = ["0.05", "8", "-0.6"]
SHIFT = ["0.9", "1", "0.007"]
SCALE = ["-0.004", "0.0625", "0.015625"]
Z = "0"


def nnet_text():
    out = ["<Nnet> "]
    for W, B, act in ((W0, B0, "<Sigmoid>"), (W1, B1, "<Sigmoid>"), (W2, B2, "<Softmax>")):
        out.append(f"<AffineTransform> {len(W)} {len(W[0])} ")
        out.append("<LearnRateCoef> 1 <BiasLearnRateCoef> 1 <MaxNorm> 0  [")
        for i, row in enumerate(W):
            out.append("  " + " ".join(row) + (" ]" if i == len(W) - 1 else " "))
        out.append(" [ " + " ".join(B) + " ]")
        out.append(f"{act} {len(W)} {len(W)} ")
    out.append("</Nnet> ")
    return "\n".join(out) + "\n"


def transform_text():
    return ("<Nnet> \n<Splice> 3 3 \n[ -5 -4 -3 -2 -1 0 1 2 3 4 5 ]\n"
            "<AddShift> 3 3 \n<LearnRateCoef> 0 [ " + " ".join(SHIFT) + " ]\n"
            "<Rescale> 3 3 \n<LearnRateCoef> 0 [ " + " ".join(SCALE) + " ]\n</Nnet> \n")


def i32(v):
    return struct.pack(">i", v)


def f32s(tokens):
    return b"".join(bytes.fromhex(HEX[t]) for t in tokens)


def layer(rows, bias):
    return i32(len(rows[0])) + i32(len(rows)) + b"".join(f32s(r) for r in rows) + f32s(bias)


def aligned_4_16():
    l0 = [W0[0] + [Z], W0[1] + [Z]] + [[Z] * 4 for _ in range(14)]
    b0 = B0 + [Z] * 14
    l1 = [W1[0] + [Z] * 14, W1[1] + [Z] * 14] + [[Z] * 16 for _ in range(14)]
    b1 = B1 + [Z] * 14
    l2 = [r + [Z] * 14 for r in W2]   # output width untouched (alignment 1), input padded to 16
    return i32(3) + layer(l0, b0) + layer(l1, b1) + layer(l2, B2) + f32s(SHIFT + [Z]) + f32s(SCALE + [Z])


def extended_5_4():
    l0 = [W0[0], W0[1], W0[0], W0[1], W0[0]]                     # node i copies node i % 2, columns unchanged
    b0 = [B0[0], B0[1], B0[0], B0[1], B0[0]]
    r0 = [W1[0][0], W1[0][1], W1[0][0], W1[0][1], W1[0][0]]     # column k copies column k % 2
    r1 = [W1[1][0], W1[1][1], W1[1][0], W1[1][1], W1[1][0]]
    l1 = [r0, r1, r0, r1, r0]
    b1 = [B1[0], B1[1], B1[0], B1[1], B1[0]]
    l2 = [r + [Z] * 3 for r in W2] + [[Z] * 5]                   # align(5, 4): zeros, not copies
    b2 = B2 + [Z]
    return i32(3) + layer(l0, b0) + layer(l1, b1) + layer(l2, b2) + f32s(SHIFT) + f32s(SCALE)


if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    open(os.path.join(HERE, "final.nnet.txt"), "w", encoding="utf-8").write(nnet_text())
    open(os.path.join(HERE, "final.feature_transform.txt"), "w", encoding="utf-8").write(transform_text())
    open(os.path.join(HERE, "aligned_4_16.bin"), "wb").write(aligned_4_16())
    open(os.path.join(HERE, "extended_5_4.bin"), "wb").write(extended_5_4())
    print("wrote", sorted(os.listdir(HERE)))
