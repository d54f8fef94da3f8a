#!/bin/bash
# one-kernel list build: timing per option set, kernel trace of the builds
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_nbr
mkdir -p "$out"
if [ "${1:-}" = "tests" ]; then
  timeout -k 10 500 python -m pytest tests/test_md_device.py -x -q -m gpu > "$out/tests.log" 2>&1
  echo "tests rc $?" | tee -a "$out/tests.log"
  tail -5 "$out/tests.log"
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o nbr --output-format csv -- python "$GRAFT_REPO_ROOT/tools/nbr_probe.py" > "$GRAFT_REPO_ROOT/$out/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
grep "per build" "$out/prof.log"
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-150 | sed -n 1,32p
