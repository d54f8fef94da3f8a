#!/bin/bash
# PMC passes for the GEMM kernels of the default bench configuration; prints per-kernel averages
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on a GPU box through gpurun (GRAFT_REPO_ROOT is unset)}" || exit 1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc$i -o p --output-format csv -- python bench.py --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 10 --warmup 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc$i/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file for set $i")
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].split("(")[0]
        if "gemm" in n or "aev_" in n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in sorted(acc):
        print(n, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in acc[n].items()}, "calls", len(next(iter(acc[n].values()))))
PY
done
