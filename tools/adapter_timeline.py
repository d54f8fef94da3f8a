"""One step of the adapter loop as the device saw it: kernels and copies in start order with the gaps between them.
python tools/adapter_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the last occurrences of the forward kernel mark steps; print the window between the 3rd-last and 2nd-last forward launches
idx = [k for k, e in enumerate(ev) if "aev_forward" in e[2]]
if len(idx) < 4:
    sys.exit("not enough steps in the trace")
for which in (-12, -3):
    a, b = idx[which], idx[which + 1]
    # start the window at the first copy before the forward kernel
    while a > 0 and ev[a - 1][0] > ev[idx[which - 1]][1] and "finish" not in ev[a - 1][2]:
        a -= 1
    t0 = ev[a][0]
    prev_end = t0
    print(f"--- step window ({(ev[b][0] - t0) / 1e3:.1f} us to the next forward launch)")
    for e in ev[a:b]:
        print(f"  +{(e[0] - t0) / 1e3:8.1f} us  gap {(e[0] - prev_end) / 1e3:7.1f}  dur {(e[1] - e[0]) / 1e3:8.1f}  {e[2]}")
        prev_end = max(prev_end, e[1])
