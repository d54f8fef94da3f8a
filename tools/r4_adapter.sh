#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_adapter
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_pair_ani_adapter.py tests/test_list_and_comm_guards.py tests/test_reference_yaml.py -x -q -m gpu > "$out/tests.log" 2>&1
echo "tests rc $?"; tail -4 "$out/tests.log"
timeout -k 10 300 python tools/adapter_path_only.py > "$out/adapter.log" 2>&1; grep "plain_ms\|reneighbour_ms\|\"hostlist\|\"devlist" "$out/adapter.log"
