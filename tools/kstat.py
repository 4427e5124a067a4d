"""Average duration (us) of the kernels whose name contains each given substring, from a rocprofv3 k_kernel_stats.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    hit = [r for r in rows if pat in r["Name"]]
    print(pat, " ".join(f"{float(r['AverageNs']) / 1e3:.1f}" for r in hit), end="  |  ")
print()
