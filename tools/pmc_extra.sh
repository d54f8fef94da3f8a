#!/bin/bash
# extra counter passes over the default hot path (instruction cache, MFMA busy cycles, wait classes); run on the GPU box:
#   tools/pmc_extra.sh "CTR1 CTR2" ["CTR3 ..." ...]  -> gpurun_out/pmc_extra/<first counter>/...  + a per-kernel summary on stdout
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 20 --warmup 3"
for c in "$@"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf "${OUT:?}/$tag"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$tag -o p --output-format csv -- python bench.py $ARGS > /dev/null 2> $OUT/$tag.err
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("ani::") and ("fused" in n or "backward_fast" in n or "forward" in n):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}  ({len(v)} launches)")
PY
