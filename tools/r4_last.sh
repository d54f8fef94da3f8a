#!/bin/bash
# after the counters are in profiles/: the default bench line with traffic, the MD loop's kernel statistics, two more suite runs
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4last; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
bash tools/kstats_md.sh r04 > $O/kstats.log 2>&1; tail -3 $O/kstats.log
for i in 1 2; do timeout -k 10 600 python -m pytest tests -m gpu -q > $O/gpu_tests_$i.log 2>&1; tail -1 $O/gpu_tests_$i.log; done
