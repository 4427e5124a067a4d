#!/usr/bin/env python3
"""Kaldi nnet1 text model + feature transform -> the scorer's .bin (the reference's
FuncTest.convert recipe: loadFromTextFile, align(4, 16), saveBinary), and Kaldi feature text ->
aligned feature .bin.

    python tools/kaldi2bin.py model  final.nnet.txt final.feature_transform.txt model.bin [--extend H O]
    python tools/kaldi2bin.py feats  feats.txt feats.bin [--count N]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fast_dnn_amd import convert as CV, formats as F  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("model")
    m.add_argument("network"), m.add_argument("transform"), m.add_argument("out")
    m.add_argument("--extend", nargs=2, type=int, metavar=("HIDDEN", "OUTPUTS"),
                   help="grow the net by circular copies first (FeedForwardNetwork.extend)")
    f = sub.add_parser("feats")
    f.add_argument("text"), f.add_argument("out")
    f.add_argument("--count", type=int, default=-1, help="serializeDataMatrix featureAmount (its off-by-one is kept)")
    a = ap.parse_args()
    if a.cmd == "model":
        net = CV.load_kaldi_nnet_text(a.network, a.transform)
        if a.extend:
            net = CV.extend(net, a.extend[0], a.extend[1])
        net = CV.align(net, 4, 16)
        F.write_model_bin(a.out, net)
        for i, l in enumerate(net.layers):
            print(f"Layer {i} neuron count = {l.in_dim}")
        print(f"Output count         = {net.layers[-1].out_dim}")
    else:
        (uid, frames), = CV.load_feature_text(a.text)[:1]
        al = CV.align_features(frames, 4)
        with open(a.out, "wb") as fh:
            fh.write(CV.feature_matrix_bytes(al, a.count))
        print(f"Input Vector Count   = {frames.shape[0]}\nInput Data Dimension = {frames.shape[1]}\n"
              f"Aligned Input Data Dimension = {al.shape[1]}")


if __name__ == "__main__":
    main()
