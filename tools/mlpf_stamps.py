"""Where a tile of the fused MLP kernel spends its cycles: run with ANI_HIP_LIB pointing at a -DABLF_STAMPS build
(tools/abl_build_mlpf.sh STAMPS -DABLF_STAMPS).  usage: python tools/mlpf_stamps.py [atoms] [arith]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
arith = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = "/tmp/stamps_ani2x.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.spatial_sort(hx.water_box(atoms, seed=12345)), cutoff=5.1, skin=2.0)
ani = ani_hip.ANI(path, 0)
ani.set_option("mlp_arith", arith)
ani.compute(inp, ago=0, eflag_atom=False, vflag=False)
lib = ani_hip.lib()
lib.ani_debug_fused_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
lib.ani_debug_fused_stamps(buf, 1)
for k in range(5):
    ani.compute(inp, ago=1, eflag_atom=False, vflag=False)
rc = lib.ani_debug_fused_stamps(buf, 0)
names = ["prologue (ring fill, constants)", "F1", "celu 1", "F2", "celu 2", "F3", "last layer + seed", "B3", "B2", "B1 + store"]
tot = sum(buf[k] for k in range(10))
print(f"stamps build: {rc == 1}; total cycles per step (sum over {atoms // 128}+ tiles, wave 0): {tot / 5:.0f}")
for k, n in enumerate(names):
    print(f"  {n:34s} {buf[k] / 5:14.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
ani.close()
