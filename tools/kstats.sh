#!/bin/bash
# per-kernel average durations of the default bench workload (rocprofv3 --kernel-trace --stats); prints the ani:: kernels
# usage: tools/kstats.sh [tag] [extra bench args]
TAG=${1:-k}; shift
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/kstats_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python bench.py --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 40 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ani::" in n and float(r["Percentage"]) > 0.3:
        print(f'{n.split("(")[0].replace("void ","")[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f}  {r["Percentage"]}%')
PY
