#!/bin/bash
# Round profile of the default bench workload (100 002-atom water, 1 member, pruned AEV; hot path + MD loop):
#   * rocprofv3 --kernel-trace --stats                      -> kernel_stats.csv
#   * separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ)  -> pmc_summary.json, tagged with the digest of the kernel
#     sources it was measured on (bench.py uses its numbers only while that digest still matches)
# Run on the GPU box from the repo root:  tools/profile_round.sh [TAG]   -> gpurun_out/profile_TAG/
TAG=${1:-r04}
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
OUT=gpurun_out/profile_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--no-cpu-baseline --no-dense-pass --no-extra --steps 40 --warmup 5"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o p --output-format csv -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$tag.err
done
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
out_dir = sys.argv[1]
kern = {}
for f in glob.glob(out_dir + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("ani::"):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    steps = max((len(v) for n, cs in acc.items() if n.startswith("ani::aev_backward") for v in cs.values()), default=1)
    for n, cs in acc.items():
        for c, v in cs.items():
            kern.setdefault(n, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v), "launches_per_step": len(v) / steps}
summary = {"source_digest": bench.source_digest(), "workload": "water-100002, 1 member, pruned AEV, bench.py default (hot path + MD loop)",
           "units": "FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE counts half of a wide read on gfx950); SQ_*_CYCLES and SQ_ACTIVE/WAIT in quad-cycles summed over waves",
           "kernels": kern}
json.dump(summary, open(out_dir + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for n in sorted(kern):
    if any(k in n for k in ("aev_", "nbr_compact", "gemm_grouped", "mlp_")):
        print(n[:60], {c: round(x["mean_per_launch"] / 1e6, 2) for c, x in sorted(kern[n].items())})
PY
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-160
