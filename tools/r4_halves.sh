#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4_halves; mkdir -p $O
timeout -k 10 400 python tools/halves_probe.py 2>&1 | grep "atoms x" | tee $O/probe.log
for n in 50001 100002; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --atoms $n > $O/bench_$n.json 2> $O/bench_$n.err; python -c "
import json;d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]);print($n, d['ms_per_step'], d['hot_path']['phase_ms'])"; done
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
