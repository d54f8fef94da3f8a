"""dE/dAEV rows of the 16-row fused MLP kernel, 128-row form against its 64-row form, many evaluations: how often and where do
they differ?  usage: [ANI_HIP_LIB=...] python tools/mlpg_debug.py [arith] [evaluations]"""
import sys, os, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

arith = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = "/tmp/dbg.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.water_box(1500, seed=5))


def handle(rows):
    ani = ani_hip.ANI(path, 0)
    ani.set_option("mlp_fused", 2)
    ani.set_option("mlp_arith", arith)
    ani.set_option("mlp_fused_gen", 1)
    ani.set_option("mlp_fused_rows", rows)
    return ani


def gaev(ani, ago):
    ani.compute(inp, ago=ago)
    v = ani.debug_view()
    return ani.debug_read(v.d_gaev, (v.nrows, v.aev_stride), np.float32).copy()


a64 = handle(64)
ref = gaev(a64, 0)
same64 = all(np.array_equal(gaev(a64, 1), ref) for _ in range(5))
a64.close()
ani = handle(128)
bad = 0
where = collections.Counter()
for k in range(nev):
    g = gaev(ani, 0 if k == 0 else 1)
    d = np.abs(g - ref).max(1)
    rows = np.nonzero(d > 0)[0]
    if len(rows):
        bad += 1
        for w in sorted(set((int(r) // 128, (int(r) % 128) // 16) for r in rows)):
            where[w] += 1
print(f"lib {os.environ.get('ANI_HIP_LIB', 'default')}: arith {arith}: kernel {ani.last_mlp_kernel()}; 64-row form repeatable: {same64}; "
      f"{bad} of {nev} evaluations differ from the 64-row form; (tile, wave) hit: {dict(where)}")
ani.close()
