#!/bin/bash
# MLP phase time of library variants (tools/abl/libani_<NAME>.so) at several sizes: usage tools/abl_run_mlpg.sh "sizes" NAME...
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
SIZES=$1; shift
for v in "$@"; do
  for n in $SIZES; do
    ANI_HIP_LIB=$PWD/tools/abl/libani_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --no-md --steps 40 --atoms $n 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['config']['atoms'], round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms'].items()})"
  done
done
