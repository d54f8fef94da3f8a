"""Print the kernel timeline (start offset, duration, name) of the last full step found in a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last pack_kernel but one -> one whole step
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:90]}")
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
