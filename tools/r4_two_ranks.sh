#!/bin/bash
# two ranks on one card over gloo: what the multi-rank loop costs around the transport
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_two
mkdir -p "$out"
export ANI_BENCH_BACKEND=gloo ANI_BENCH_NATIVE_COMM=try ANI_COMM_DISABLE_RCCL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python bench.py --gpus 2 --steps 90 --warmup 10 --no-cpu-baseline --no-dense-pass --no-extra ${BENCH_ARGS:-} > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_two/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d["config"])
print(json.dumps(d["md_loop"], indent=1)[:2500])
PY
