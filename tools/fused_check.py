"""fused MLP against the per-layer kernels on the same inputs: python tools/fused_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

for kind, nm, box in (("ani2x", 1, hx.water_box(1500, seed=5)), ("ani1x", 1, hx.random_box(64, 4, 9.0, seed=3)),
                      ("ani1x", 2, hx.random_box(64, 4, 9.0, seed=3)), ("ani2x", 2, hx.random_box(700, 7, 22.0, seed=9))):
    path = f"/tmp/fc_{kind}_{nm}.anim"
    mf.write_model(path, mf.synthetic_model(kind, nm, seed=2024))
    inp = hx.decompose(box)
    out = {}
    for fused in (0, 2):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused", fused)
        out[fused] = ani.compute(inp, ago=0)
        ani.close()
    df = np.abs(out[2]["force"] - out[0]["force"])
    de = np.abs(out[2]["eatom"] - out[0]["eatom"])
    sp = np.asarray(inp.species[: inp.nlocal])
    print(f"{kind} x{nm}: |dE| {abs(out[2]['energy'] - out[0]['energy']):.3e}  max|dF| {df.max():.3e}  max|de_atom| {de.max():.3e}  "
          + "  ".join(f"s{s}:{de[sp == s].max():.1e}" for s in np.unique(sp)), flush=True)
