"""Fixture for tests/test_convert.py from the reference's own data files (run where
/root/reference exists):  python tests/golden/make_convert_golden.py

feat_text_head.npz holds the first rows of data/16khz (Kaldi feature TEXT, the reference's
input) and the same rows of data/16khz.bin (what the reference's Java tooling made of them:
BatchData.loadFromText -> alignDimension(4) -> serializeDataMatrix), plus both headers.
"""
import os
import struct

import numpy as np

REF = "/root/reference/data"
ROWS = 6
out = {}
for name, binname in (("16khz", "16khz.bin"), ("8khz", "8khz.aligned.bin")):
    with open(os.path.join(REF, name), "r", encoding="utf-8") as fh:
        lines = fh.read().splitlines()
    head = "\n".join(lines[: 1 + ROWS]) + " ]\n"  # "utt [" + ROWS rows, closed
    raw = open(os.path.join(REF, binname), "rb").read()
    n, dim = struct.unpack(">ii", raw[:8])
    body = np.frombuffer(raw[8:], dtype=">f4").reshape(-1, dim)
    out[f"{name}_text_head"] = np.array(head)
    out[f"{name}_bin_rows"] = body[:ROWS].astype(np.float32)
    out[f"{name}_header"] = np.array([n, dim, body.shape[0]], dtype=np.int64)  # header n, dim, rows actually in the file
np.savez_compressed(os.path.join(os.path.dirname(__file__), "feat_text_head.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
