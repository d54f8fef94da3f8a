#!/bin/bash
# PMC counters of the MLP kernels in the hot path of the default bench workload.  usage: tools/pmc_mlpf.sh TAG ["CTRS" ...]
TAG=${1:?usage: TAG [counter groups]}; shift
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 20 --warmup 3"
n=0
for grp in "$@"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$n -o p --output-format csv -- python bench.py $ARGS > /dev/null 2> $OUT/err$n.log
done
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("ani::"):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(n, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for n in sorted(out):
    if any(k in n for k in ("mlp_", "gemm_")):
        print(n[:70])
        for c, x in sorted(out[n].items()):
            print(f"    {c:28s} {x['mean_per_launch']/1e6:12.3f} M")
PY
