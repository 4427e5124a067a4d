#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE ITSELF.

Runs only where /root/reference exists: it drives oracle/_ref/libfastdnn_ref.so
(the reference's dnn.cc / float_dnn.cc compiled where they lie, canonical flags
-O2 -msse4 -ffp-contract=off; see oracle/Makefile + oracle/ref_tap.cpp) and
stores inputs + the reference's outputs / intermediate state as small .npz
fixtures.  The fixtures are data only; no reference source travels.

    python tests/golden/make_golden.py

Fixtures
  lut.npz          QuantizedSigmoid table (1280 bytes) + get() probes
  quantizer.npz    QuantizedSimdLayer edge cases (cut-off, wrap to -128, 127/0)
  tiny.npz         432->3x64->100 net (.bin bytes inside), 100 frames of the
                   shipped data/16khz.bin, every per-layer tap
  mid_lazy.npz     432->3x256->1000 (seed-regenerated, sha256 pinned), 40 % masks
                   with 3 % churn, LazyContext outputs + dense outputs
  net_full.npz     432->7x2048->8000 (seed-regenerated, sha256 pinned), 16 frames:
                   sha256 of every layer's u8 activations / accumulators, full
                   soft-max rows for 4 frames, saturation-event count
  sat.npz          small net driven into pmaddubsw saturation (features x50)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fast_dnn_amd import formats as F  # noqa: E402
from oracle.oracle import Oracle, RefLib  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TMP = os.environ.get("TMPDIR", "/tmp")
REF_DATA = "/root/reference/data"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main() -> None:
    ref = RefLib()
    ref_fma = RefLib(fma=True)
    x16 = F.read_feature_bin(os.path.join(REF_DATA, "16khz.bin"))  # 100 x 432
    x8 = F.read_feature_bin(os.path.join(REF_DATA, "8khz.aligned.bin"))[:100]

    # ---- (iv) LUT + get() probes
    probes = np.array([-1e9, -7, -6.405, -6.4, -6.395, -6.39, -0.015, -0.005, -0.0049999, 0, 0.0049999, 0.005, 0.015,
                       0.025, 1.0, 2.5, 6.39, 6.395, 6.4, 6.405, 100, 1e9], dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(99))
    probes = np.concatenate([probes, (rng.standard_normal(4000) * 3).astype(np.float32)])
    np.savez_compressed(os.path.join(OUT, "lut.npz"), lut=ref.lut(), probes=probes,
                        probe_out=np.array([ref.sigmoid_get(float(p)) for p in probes], dtype=np.uint8))

    # ---- (v) quantizer edge cases
    q = {}
    cases = []
    w = (rng.standard_normal((32, 64)) * 0.05).astype(np.float32)
    cases.append(("plain", w, 3.0))
    w = rng.standard_normal((32, 64)).astype(np.float32)
    w[3, 5] = 7.0
    w[4, 6] = -9.0
    cases.append(("over_cutoff", w, 3.0))  # upper clamp is dead code: 7*42 wraps mod 256
    w = (rng.standard_normal((16, 32)) * 0.3).astype(np.float32)
    w[0, 0] = np.float32(2.9882)
    cases.append(("round_to_128", w, 3.0))  # 2.9882*43 = 128.49 -> (char)128 = -128
    w = np.zeros((16, 16), dtype=np.float32)
    cases.append(("all_zero", w, 3.0))  # 127/0 = inf, 0*inf = NaN -> 0
    w = (rng.standard_normal((16, 48)) * 2.0).astype(np.float32)
    cases.append(("cutoff_1", w, 1.0))
    w = np.zeros((16, 16), dtype=np.float32)
    w[0, 0] = np.float32(1.0538998)
    w[1, 1] = -np.float32(1.0538998)
    cases.append(("wrap_pm", w, 3.0))
    for name, w, cut in cases:
        wq, mult = ref.quantize(w, cut)
        q[f"{name}_w"] = w
        q[f"{name}_cut"] = np.float32(cut)
        q[f"{name}_wq"] = wq
        q[f"{name}_mult"] = np.float32(mult)
    q["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "quantizer.npz"), **q)

    # ---- (i) tiny net, every tap, both shipped feature files
    tiny_path = os.path.join(TMP, "golden_tiny.bin")
    F.write_model_bin(tiny_path, F.synth_net([432, 64, 64, 64, 100], seed=3))
    r = ref.load(tiny_path)
    t16 = r.forward_taps(x16, batch=10)
    assert (r.calculate(x16, batch=10) == t16["probs"]).all()
    assert (r.calculate(x16, batch=7) == t16["probs"]).all()  # blocking never changes results
    assert (r.hidden_acts(x16, batch=8) == t16["u8_acts"][-1]).all()
    p8 = r.calculate(x8, batch=10)
    rf = ref_fma.load(tiny_path)
    tf = rf.forward_taps(x16, batch=10)
    np.savez_compressed(
        os.path.join(OUT, "tiny.npz"),
        model_bin=np.frombuffer(open(tiny_path, "rb").read(), dtype=np.uint8),
        x16=x16, x8=x8, probs8=p8,
        wq=np.concatenate([r.q_weights(j).ravel() for j in range(r.n_q)]),
        mult=np.array([r.q_mult(j) for j in range(r.n_q)], dtype=np.float32),
        l0_lin=t16["l0_lin"], u8_acts=t16["u8_acts"], acc_hid=t16["acc_hid"].astype(np.int32),
        acc_out=t16["acc_out"].astype(np.int32), logits=t16["logits"], probs=t16["probs"],
        fma_l0_lin=tf["l0_lin"], fma_u8_acts=tf["u8_acts"], fma_probs=tf["probs"],
    )
    assert (t16["acc_hid"].astype(np.int32).astype(np.float32) == t16["acc_hid"]).all()
    r.close()
    rf.close()

    # ---- (ii) mid net, lazy path
    mid_path = os.path.join(TMP, "golden_mid.bin")
    F.write_model_bin(mid_path, F.synth_net([432, 256, 256, 256, 1000], seed=5))
    r = ref.load(mid_path)
    masks = F.generate_masks(100, 1000, ratio=0.40, churn=0.03, seed=11)
    lazy = r.lazy(x16, masks, batch=8)
    dense = r.calculate(x16, batch=10)
    np.savez_compressed(os.path.join(OUT, "mid_lazy.npz"), model_sha256=F.sha256_file(mid_path), masks_sha256=sha(masks),
                        mask_rows=masks[:3], lazy=lazy[:40], dense=dense[:20],
                        hidden_last=r.hidden_acts(x16, batch=8))
    r.close()

    # ---- saturation fixture: small net pushed into pmaddubsw saturation
    sat_path = os.path.join(TMP, "golden_sat.bin")
    F.write_model_bin(sat_path, F.synth_net([432, 128, 128, 128, 200], seed=8))
    r = ref.load(sat_path)
    xs = (x16[:32] * 50).astype(np.float32)
    ts = r.forward_taps(xs, batch=10)
    o = Oracle(sat_path)
    _, ot = o.calculate(xs, taps=True)
    assert ot["sat_events"] > 0
    np.savez_compressed(os.path.join(OUT, "sat.npz"), model_sha256=F.sha256_file(sat_path), x=xs,
                        u8_acts=ts["u8_acts"], acc_hid=ts["acc_hid"].astype(np.int32), acc_out=ts["acc_out"].astype(np.int32),
                        probs=ts["probs"], sat_events=np.int64(ot["sat_events"]))
    r.close()

    # ---- (iii) full-size net
    net_path = os.path.join(TMP, "golden_net.bin")
    F.ensure_model_file(net_path, F.NET_TOPOLOGY, seed=1, mode="gauss")
    r = ref.load(net_path)
    xn = x16[:16]
    tn = r.forward_taps(xn, batch=8)
    o = Oracle(net_path)
    po, ot = o.calculate(xn, batch=8, taps=True)
    np.savez_compressed(
        os.path.join(OUT, "net_full.npz"),
        model_sha256=F.sha256_file(net_path), model_size=np.int64(os.path.getsize(net_path)),
        x=xn, mult=np.array([r.q_mult(j) for j in range(r.n_q)], dtype=np.float32),
        wq_sha256=np.array([sha(r.q_weights(j)) for j in range(r.n_q)]),
        u8_sha256=np.array([sha(tn["u8_acts"][j]) for j in range(tn["u8_acts"].shape[0])]),
        acc_hid_sha256=np.array([sha(tn["acc_hid"][j].astype(np.int32)) for j in range(tn["acc_hid"].shape[0])]),
        acc_out_sha256=sha(tn["acc_out"].astype(np.int32)),
        probs4=tn["probs"][:4], probs_sha256=sha(tn["probs"]),
        sat_events=np.int64(ot["sat_events"]),
        risky_pairs=np.array([o.risky_pairs(j) for j in range(1, o.n_layers)], dtype=np.int64),
    )
    assert (po == tn["probs"]).all(), "oracle != reference on the full-size net"
    r.close()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f:16s} {os.path.getsize(os.path.join(OUT, f)):8d} B")


if __name__ == "__main__":
    main()
