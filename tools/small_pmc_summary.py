#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (the small / mid-size call profiles of tools/profile_small.sh).

    python tools/small_pmc_summary.py out.json pass1.csv pass2.csv ...

Kernels are told apart by their full template name (qgemm_small_kernel<1,...> = hidden layer, <2, true, ...> = output layer).
The second half of each kernel's dispatches is averaged (the first half warms the caches and the clocks).  FETCH_SIZE is doubled
as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests are tallied at 64 bytes); bytes are per launch."""
import csv, json, sys, collections


def short(k: str) -> str:
    k = k.replace("void ", "").replace("fdnn::(anonymous namespace)::", "").replace("fdnn::", "")
    depth, out = 0, []
    for ch in k:  # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception as e:  # noqa: BLE001
        print("missing", f, e)
        continue
    for r in rows:
        k = r["Kernel_Name"]
        if "fdnn" not in k or "fastdiv" in k or "xor80" in k or "image" in k:
            continue
        acc[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, c in acc.items():
    d = {k: round(sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])), 1) for k, v in c.items()}
    d["dispatches"] = max(len(v) for v in c.values())
    if "FETCH_SIZE" in d:
        d["hbm_side_read_bytes"] = int(2 * d["FETCH_SIZE"] * 1000)
    if "WRITE_SIZE" in d:
        d["hbm_side_write_bytes"] = int(d["WRITE_SIZE"] * 1000)
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit"] = round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        d["mfma_busy_of_sq_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, d["SQ_BUSY_CYCLES"] * 4), 4)  # (busy cycles are per SE-quad: x4 SIMDs)
    if "SQ_WAVE_CYCLES" in d:
        for k_ in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k_ in d:
                d[k_.lower() + "_of_wave_cycles"] = round(d[k_] / max(1.0, d["SQ_WAVE_CYCLES"]), 3)
    out[n] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
for n, d in out.items():
    print(n, {k: d[k] for k in d if k.endswith("_of_wave_cycles") or k in ("l2_hit", "hbm_side_read_bytes", "hbm_side_write_bytes", "dispatches")})
