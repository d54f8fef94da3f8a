#!/bin/bash
# round-4 evidence in one call: GPU suite, rocprofv3 stats + PMC passes (tied to the source digest) installed under profiles/ on the box,
# then the default bench line (with counter traffic), the per-GPU-share sizes, the short window, the long run, the NVE run, the
# eight-member boxes and the MD loop's kernel statistics
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4ev; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
cp gpurun_out/profile_r04/pmc_summary.json profiles/r04_pmc_summary.json
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
for n in 12501 25002 50001; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --atoms $n > $O/bench_$n.json 2> $O/bench_$n.err; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> /dev/null
timeout -k 10 300 python tools/long_run.py > $O/md_long_run.json 2> $O/md_long_run.err
timeout -k 10 300 python tools/md_probe.py 100002 1000 0.25 1 300 > $O/md_nve_100k_1000steps.log 2>&1
timeout -k 10 300 python tools/members_probe.py "" > $O/members_probe.log 2>&1
bash tools/kstats_md.sh r04 > $O/kstats.log 2>&1; tail -1 $O/kstats.log
rm -f gpurun_out/kstats_r04/s_kernel_trace.csv gpurun_out/profile_r04/*/*kernel_trace.csv gpurun_out/profile_r04/*/*counter_collection.csv
echo done
