#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into one per-kernel JSON summary.

    python tools/pmc_summary.py out.json pass1.csv pass2.csv ...

Per kernel (our four hot kernels only) the counters are averaged over dispatches (the first
three dispatches of each kernel are warm-up and dropped).  Derived values follow
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 kB
per count here and FETCH_SIZE is doubled on gfx950 (64-B request granularity is reported at
half size); bytes are per launch.
"""
import csv
import json
import sys
from collections import defaultdict

NAMES = {
    "l0_valu_kernel": "l0_valu_kernel<4,16> (fp32 layer 0, canonical flavour, small batches)",
    "l0_chain_kernel": "l0_chain_kernel (fp32 layer 0, canonical flavour, one chain per pass)",
    "l0_image_kernel": "l0_image_kernel (layer-0 frame image: shift/scale + chain-major transpose)",
    "l0_mfma_kernel": "l0_mfma_kernel (fp32 layer 0 on the matrix pipe: screened canonical / fused flavour)",
    "l0_split_kernel": "l0_split_kernel (int8 screening of layer 0)",
    "l0_digits_kernel": "l0_digits_kernel (frames -> int8 digit planes + row constants)",
    "l0_fix_list_kernel": "l0_fix_list_kernel (exact recomputation of the flagged outputs, one list per launch)",
    "l0_fix_kernel": "l0_fix_kernel (exact recomputation of the screened outputs)",
    "l0_xnorm_kernel": "l0_xnorm_kernel (frame norms)",
    "normalize_kernel": "normalize_kernel (soft-max scale)",
}


def label(kernel_name: str):
    if "qchain_kernel" in kernel_name:
        args = kernel_name.split("qchain_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return f"qchain_kernel (all hidden layers in one launch, 256x{32 * int(args[0]) * int(args[1])} tile, {4 * int(args[1])} waves, BK{args[2]})"
    if "qppo_kernel" in kernel_name:
        return "qppo_kernel output (role-split fused output layer, 256x160 halves, 8 waves, BK128)"
    if "qgemm_kernel" in kernel_name:
        args = kernel_name.split("qgemm_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        kind = "output" if args[4] == "true" else "hidden"
        ft = 32 * int(args[0]) * int(args[1])
        return f"qgemm_kernel {kind} (256x{ft} tile, {4 * int(args[1])} waves, BK{args[2]})"
    for k, v in NAMES.items():
        if k in kernel_name:
            return v
    return None


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        seen = defaultdict(int)
        try:
            rows = list(csv.DictReader(open(p)))
        except OSError as e:
            print("skip", p, e)
            continue
        per_dispatch = defaultdict(dict)
        for r in rows:
            lab = label(r["Kernel_Name"])
            if lab:
                per_dispatch[(int(r["Dispatch_Id"]), lab)][r["Counter_Name"]] = float(r["Counter_Value"])
        for (did, lab), ctrs in sorted(per_dispatch.items()):
            seen[lab] += 1
            if seen[lab] <= 3:
                continue
            for c, v in ctrs.items():
                acc[lab][c].append(v)
    res = {}
    for lab, ctrs in acc.items():
        d = {c: round(sum(v) / len(v)) for c, v in ctrs.items()}
        der = {}
        if "GRBM_GUI_ACTIVE" in d:
            der["kernel_cycles_per_xcd"] = round(d["GRBM_GUI_ACTIVE"] / 8)
        if "FETCH_SIZE" in d:
            der["hbm_read_MB_per_launch (FETCH_SIZE KB x2 gfx950 correction)"] = round(2 * d["FETCH_SIZE"] / 1000, 1)
        if "WRITE_SIZE" in d:
            der["hbm_write_MB_per_launch (WRITE_SIZE KB)"] = round(d["WRITE_SIZE"] / 1000, 1)
        if "TCC_HIT_sum" in d and d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0):
            der["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"]:
            der["mfma_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))"] = round(
                d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
            w = d["SQ_WAVE_CYCLES"]
            wait = d.get("SQ_WAIT_ANY", 0)
            active = d.get("SQ_ACTIVE_INST_ANY", 0)
            der["wave_time_split (active / issue-stall / waitcnt+barrier)"] = [
                round(active / w, 2), round(max(0.0, 1 - active / w - wait / w), 2), round(wait / w, 2)]
        d["derived"] = der
        res[lab] = d
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
