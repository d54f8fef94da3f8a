#!/bin/bash
# round-4 first measurements: MFMA shadow probe, timeline of one MD step at 12 501 atoms, today's baseline numbers
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r4a
timeout -k 10 120 ./tools/abl/mfma_shadow_probe > gpurun_out/r4a/shadow_probe.log 2>&1
OUT=gpurun_out/r4a/trace12k
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o s --output-format csv -- python bench.py --no-cpu-baseline --no-dense-pass --no-extra --steps 60 --warmup 10 --atoms 12501 > "$OUT/bench.json" 2> "$OUT/err.log"
python tools/timeline.py "$(find $OUT -name '*kernel_trace.csv' | head -1)" > gpurun_out/r4a/timeline_12501.log 2>&1
rm -f $(find $OUT -name '*kernel_trace.csv')
for n in 12501 25002 50001; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --atoms $n > gpurun_out/r4a/bench_$n.json 2> gpurun_out/r4a/bench_$n.err
done
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
echo done
