#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4_halves; mkdir -p $O
ANI_FUSED_AUTOTUNE_VERBOSE=1 timeout -k 10 400 python tools/halves_probe.py 2>&1 | grep "atoms x\|libani_hip: fused" | tee $O/probe.log
timeout -k 10 600 python -m pytest tests/test_mlp_fused.py tests/test_baseline_workloads.py -q -m gpu > $O/tests.log 2>&1; tail -1 $O/tests.log
for a in "--atoms 50001" "--atoms 37500" "--atoms 75000"; do ANI_FUSED_AUTOTUNE_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-dense-pass --no-extra $a > $O/b.json 2> $O/b.err; grep "libani_hip: fused" $O/b.err | head -2; python -c "
import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$a', d['ms_per_step'], d['hot_path']['phase_ms']['mlp'])"; done
ANI_FUSED_AUTOTUNE_VERBOSE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-dense-pass > $O/bfull.json 2> $O/bfull.err; grep "libani_hip: fused" $O/bfull.err | head -5; python -c "
import json;d=json.loads(open('$O/bfull.json').read().strip().splitlines()[-1]);print('default', d['ms_per_step'], d['hot_path']['phase_ms']['mlp'], 'extra', d['extra_config']['ms_per_step'], d['extra_config']['phase_ms'], 'mixed', d['mixed_species_config']['ms_per_step'])"
