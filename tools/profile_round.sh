#!/bin/bash
# Round profile of the default bench workload (100 002-atom water, 1 member, pruned AEV): kernel stats + HBM counters.
# Run on the GPU box from the repo root; writes under gpurun_out/profile_round/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profile_round
rm -rf $OUT && mkdir -p $OUT
ARGS="--no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 40 --warmup 5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python bench.py $ARGS > $OUT/bench_stats.json 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o p --output-format csv -- python bench.py $ARGS > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/profile_round/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("ani::"):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(n, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open("gpurun_out/profile_round/pmc_summary.json", "w"), indent=1, sort_keys=True)
for n in sorted(out):
    print(n, {c: round(x["mean_per_launch"], 1) for c, x in out[n].items()})
PY
cp $(find $OUT/stats -name "*kernel_stats.csv") $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-150
