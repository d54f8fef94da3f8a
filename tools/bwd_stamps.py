"""per-phase cycle counts of the AEV backward kernel (a -DABLB_STAMPS build: tools/abl_build.sh STAMPS "-DABLB_STAMPS"):
ANI_HIP_LIB=tools/abl/libani_STAMPS.so python tools/bwd_stamps.py [atoms]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
path = "/tmp/stamps_ani2x.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.spatial_sort(hx.water_box(atoms, seed=12345)))
ani = ani_hip.ANI(path, 0)
ani.compute(inp, ago=0, eflag_atom=False, vflag=False)
lib = ani_hip.lib()
lib.ani_debug_fused_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
lib.ani_debug_fused_stamps(buf, 1)
N = 5
for k in range(N):
    ani.compute(inp, ago=1, eflag_atom=False, vflag=False)
rc = lib.ani_debug_fused_stamps(buf, 0)
names = ["header load (drains the wave's outstanding atomics)", "list + dE/dAEV row loads", "radial stage + radial-only scatter",
         "row table", "angular stage", "final scatter + centre force"]
tot = sum(buf[k] for k in range(6))
nc = max(buf[8], 1)
print(f"stamps build: {rc == 2}; centres stamped (wave 0 of each workgroup, {N} steps): {nc}; cycles per centre: {tot / nc:.0f}")
for k, n in enumerate(names):
    print(f"  {n:55s} {buf[k] / nc:9.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
fn = ["header load", "list loads", "unpack lists into LDS (+ 2 cos per neighbour)", "radial", "bucket table", "angular (phase 1 + 2 per chunk)", "row store"]
ftot = sum(buf[16 + k] for k in range(7))
fc = max(buf[24], 1)
print(f"forward kernel: centres stamped {fc}; cycles per centre: {ftot / fc:.0f}")
for k, n in enumerate(fn):
    print(f"  {n:55s} {buf[16 + k] / fc:9.0f}  {100.0 * buf[16 + k] / max(ftot, 1):5.1f} %")
ani.close()
