#!/usr/bin/env python3
"""Quantization error of the scorer against the fp32 net (the reference's FuncTest.diff notion:
per output node, |quantized - float| summed over the frames; nodes above 0.1 get printed there).

    python tools/accuracy_report.py model.bin [features.bin] [--frames N]      (needs the GPU)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fast_dnn_amd import api, convert as CV, formats as F  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("features", nargs="?")
    ap.add_argument("--frames", type=int, default=1000)
    a = ap.parse_args()
    net = F.read_model_bin(a.model)
    x = F.read_feature_bin(a.features) if a.features else F.synth_features(a.frames, net.layers[0].in_dim, seed=3)
    x = x[: a.frames]
    dnn = api.QuantizedDnn.loadFromFile(a.model)
    q = dnn.calculate(x)
    dnn.delete()
    print(json.dumps(CV.quantization_report(CV.float_forward(net, x), q), indent=1))


if __name__ == "__main__":
    main()
