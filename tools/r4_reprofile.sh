#!/bin/bash
# after a change under csrc/: the GPU suite, the rocprofv3 stats + PMC passes (profiles are tied to the source digest), then the default bench line
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4re; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log
cp gpurun_out/profile_r04/pmc_summary.json profiles/r04_pmc_summary.json
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json
