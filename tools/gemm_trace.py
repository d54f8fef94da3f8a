"""Per-launch durations of the six grouped GEMMs of a step from a rocprofv3 --kernel-trace CSV (argv[1])."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq, per = [], collections.defaultdict(list)
pos = 0
for r in rows:
    n = r["Kernel_Name"]
    if "aev_forward" in n:
        pos = 0
    if "gemm_grouped" in n:
        per[(pos, n.split("(")[0])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        pos += 1
for k in sorted(per):
    v = sorted(per[k])[len(per[k]) // 2]
    print(k, "calls", len(per[k]), "median_us", v / 1e3)
