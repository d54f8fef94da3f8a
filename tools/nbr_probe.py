"""Time of ani_build_list_device at the headline size, per option set (GPU).  python tools/nbr_probe.py [natoms]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402

_pkg.load()
import torch  # noqa: E402
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf  # noqa: E402

natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
path = "/tmp/nbr_probe.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1))
sysm = hx.spatial_sort(hx.water_box(natoms))
inp = hx.decompose(sysm)
dev = torch.device("cuda:0")
x = torch.as_tensor(inp.x, dtype=torch.float64, device=dev).contiguous()
sp = torch.as_tensor(inp.species.astype(np.int32), device=dev)
cut = 7.1
lo, hi = inp.x.min(0) - 0.25, inp.x.max(0) + 0.25
for name, opts in (("search + sort kernels", dict(nbr_sorted_rows=0)), ("one kernel", dict(nbr_sorted_rows=1, nbr_half_cells=0)),
                   ("one kernel, half cells", dict(nbr_sorted_rows=1, nbr_half_cells=1))):
    ani = ani_hip.ANI(path, 0)
    for k, v in opts.items():
        ani.set_option(k, v)
    for _ in range(3):
        n = ani.build_list_device(inp.ntotal, inp.nlocal, sp.data_ptr(), x.data_ptr(), cut, lo, hi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        n = ani.build_list_device(inp.ntotal, inp.nlocal, sp.data_ptr(), x.data_ptr(), cut, lo, hi)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:28s} {dt * 1e3:.3f} ms per build  ({n} pairs, ntotal {inp.ntotal})", flush=True)
    ani.close()
