#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4_split; mkdir -p $O
for sp in "0,0" "0,40" "0,54" "0,80" "0,108" "0,216" "20,40" "40,0" "84,0" "168,0"; do
  MEMBERS_PROBE_FIRST_ONLY=1 ANI_FUSED_SPLIT="$sp" timeout -k 10 120 python tools/members_probe.py "" 2>&1 | grep "water-10002" | sed "s/^/split $sp: /"
done | tee $O/split_10002x8.log
