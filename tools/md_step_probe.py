"""Development probe: per-step wall time of the device MD loop (synchronised every step), split into steps that
re-neighbour and steps that do not.  usage: python tools/md_probe.py [atoms] [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, comm, harness as hx, md, model_file as mf

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
system = hx.spatial_sort(hx.water_box(atoms))
path = "/tmp/md_probe.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024, out_scale=0.02))
inp = hx.decompose(system, (1, 1, 1), 0, cutoff=5.1, skin=2.0)
ani = ani_hip.ANI(path, 0, -1, use_cuaev=True, use_fullnbr=True, use_single=True)
run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo, grid=(1, 1, 1))
run.create_velocities(300.0)
for _ in range(10):
    run.step()
torch.cuda.synchronize()
normal, rebuild = [], []
t_all = time.perf_counter()
for _ in range(steps):
    b = run.nbuilds
    t0 = time.perf_counter()
    run.step()
    torch.cuda.synchronize()
    (rebuild if run.nbuilds > b else normal).append(time.perf_counter() - t0)
t_all = time.perf_counter() - t_all
print(f"atoms {atoms}: {steps} steps, {len(rebuild)} rebuilds; plain step {np.median(normal) * 1e3:.3f} ms (median, synchronised), "
      f"rebuild step {np.median(rebuild) * 1e3 if rebuild else float('nan'):.3f} ms; loop average {t_all / steps * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(steps):
    run.step()
torch.cuda.synchronize()
print(f"unsynchronised loop: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")

# where a re-neighbouring step spends its time (each piece synchronised)
import collections
acc = collections.defaultdict(list)
def timed(obj, name, key):
    fn = getattr(obj, name)
    def wrap(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[key].append(time.perf_counter() - t)
        return out
    setattr(obj, name, wrap)
timed(run.dc, "exchange", "exchange"); timed(run.dc, "borders", "borders"); timed(ani, "build_list_device", "build_list_device")
timed(run, "_per_atom_factors", "per_atom_factors"); timed(run, "_build_list", "build_list_total")
orig_forces = run._forces
def forces():
    first = run.since_build == 0
    torch.cuda.synchronize(); t = time.perf_counter()
    orig_forces()
    torch.cuda.synchronize(); acc["forces_after_rebuild" if first else "forces_plain"].append(time.perf_counter() - t)
run._forces = forces
for _ in range(steps):
    run.step()
for k, v in acc.items():
    print(f"  {k:24s} n={len(v):4d} median {np.median(v) * 1e3:.3f} ms")
