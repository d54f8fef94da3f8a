#!/bin/bash
# row tickets below the 40 000-row threshold: hot-path phases at 12 501 / 25 002 / 37 500 atoms with aev_tickets_min 40000 (default) / 10000
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r4_tickets; mkdir -p $O
for n in 12501 25002 37500; do for t in 40000 10000; do
  ANI_BENCH_OPTIONS="aev_tickets_min=$t" timeout -k 10 200 python bench.py --no-cpu-baseline --no-dense-pass --no-extra --atoms $n > $O/b_${n}_$t.json 2> $O/err.log
  python -c "
import json;d=json.loads(open('$O/b_${n}_$t.json').read().strip().splitlines()[-1]);p=d['hot_path']['phase_ms'];print($n, 'tickets_min', $t, 'step', round(d['ms_per_step'],4), 'hot', round(d['hot_path']['ms_per_step'],4), 'fwd', round(p['aev_fwd'],4), 'bwd', round(p['aev_bwd'],4), 'mlp', round(p['mlp'],4))"
done; done
