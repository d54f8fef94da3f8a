#!/bin/bash
# durations of the individual MLP launches of one step (rocprofv3 --kernel-trace), default bench workload
# usage: tools/ktrace_mlp.sh TAG [bench args]
TAG=${1:-m}; shift
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/ktrace_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o s --output-format csv -- python bench.py --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.log
python - "$(find $OUT -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ani::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: from the last pack_kernel on
idx = max(i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"])
prev = max(i for i, r in enumerate(rows[:idx]) if "pack_kernel" in r["Kernel_Name"])
t0 = int(rows[prev]["Start_Timestamp"])
for r in rows[prev:idx]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r["Grid_Size"]:>8s} wg {r["Workgroup_Size"]:>4s}  {r["Kernel_Name"].replace("void ", "")[:90]}')
PY
