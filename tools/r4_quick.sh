#!/bin/bash
# quick look at the MD loop numbers of the default bench (no CPU baseline, no extras)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_quick
mkdir -p "$out"
timeout -k 10 500 python bench.py --no-cpu-baseline --no-dense-pass --no-extra ${BENCH_ARGS:-} > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_quick/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(json.dumps(d["md_loop"], indent=1)[:1800])
PY
