#!/bin/bash
# every kernel of the bench's MD loop with calls and average duration (rocprofv3 --kernel-trace --stats), at any size
# usage: tools/kstats_md.sh TAG [bench args, e.g. --atoms 12501]
TAG=${1:-md}; shift
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/kstats_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python bench.py --no-cpu-baseline --no-dense-pass --no-extra --steps 200 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python - "$OUT/kernel_stats.csv" "$OUT/bench.json" <<'PY'
import csv, sys, json
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    print(f'{n.split("(")[0].replace("void ","")[:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
d = json.loads(open(sys.argv[2]).read())
print("md_loop", d["md_loop"]["ms_per_step"], "hot", d["hot_path"]["ms_per_step"], d["hot_path"]["phase_ms"])
PY
