"""where a re-neighbouring step of the resident loop spends its time (one rank): every section of md.VerletRun._build_list
bracketed by device synchronisations, median of N forced rebuilds; then the same rebuild unbracketed.
    python tools/rebuild_probe.py [atoms] [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, md, model_file as mf

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7
path = "/tmp/rebuild_probe.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
dev = torch.device("cuda:0")
sysm = hx.spatial_sort(hx.water_box(atoms))
inp = hx.decompose(sysm)
ani = ani_hip.ANI(path, 0)
run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=0.5, box_lo=sysm.boxlo, langevin=(300.0, 100.0), seed=7)
run.create_velocities(300.0)
for _ in range(30):
    run.step()
sync = torch.cuda.synchronize


def timed(fn):
    sync(); t = time.perf_counter(); r = fn(); sync()
    return r, (time.perf_counter() - t) * 1e3


acc = {}
def add(k, v): acc.setdefault(k, []).append(v)

for _ in range(N):
    for _ in range(5):
        run.step()
    n = run.nlocal
    (xo, v, tag, sp), t = timed(lambda: run.dc.exchange(run.x[:n], run.v, run.tag, run.species[:n])); add("exchange (wrap)", t)
    run.nlocal = n = xo.shape[0]
    run.v, run.tag = v.contiguous(), tag.contiguous()
    (x, s), t = timed(lambda: run.dc.borders(xo, sp.contiguous())); add("borders (ghost shell)", t)
    run.x, run.species = x, s
    run.ntotal = run.x.shape[0]
    _, t = timed(run._per_atom_factors); add("per-atom factors", t)
    def zf():
        run.f = torch.zeros((run.ntotal, 3), dtype=torch.float64, device=dev)
    _, t = timed(zf); add("force buffer", t)
    lo = run.dc.sub_lo - run.cutneigh - 0.25
    hi = run.dc.sub_hi + run.cutneigh + 0.25
    def bl():
        run.npairs = ani.build_list_device(run.ntotal, n, run.species.data_ptr(), run.x.data_ptr(), run.cutneigh, lo, hi, stream=run._stream)
    _, t = timed(bl); add("ani_build_list (device list)", t)
    def cl():
        run.x_built = run.x[:n].clone(); run._d2max.zero_()
    _, t = timed(cl); add("x_built copy", t)
    run.since_build = 0
    _, t = timed(run._forces); add("first force evaluation on the new list", t)
    run._final_integrate()
    _, t = timed(lambda: run.step()); add("(a plain step)", t)
    _, t = timed(lambda: run.step(force_rebuild=True)); add("(a whole re-neighbouring step, unbracketed)", t)
for k, v in acc.items():
    print(f"{k:48s} {np.median(v):8.3f} ms   (min {min(v):.3f}, max {max(v):.3f})")
ani.close()
