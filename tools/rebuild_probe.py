"""Development probe: where a re-neighbouring step of md.VerletRun spends its time (100 002-atom water box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
import numpy as np, torch
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf, md

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
dev = torch.device("cuda", 0)
system = hx.spatial_sort(hx.water_box(atoms, seed=12345))
path = "/tmp/probe_md.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024, out_scale=0.02))
inp = hx.decompose(system, (1, 1, 1), 0, cutoff=5.1, skin=2.0)
ani = ani_hip.ANI(path, 0, -1, True, True, True)
run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo)
run.create_velocities(300.0)
for _ in range(20):
    run.step()


def timed(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"step (no rebuild)      {timed(run.step, 9):.3f} ms")
print(f"_regenerate_ghosts     {timed(run._regenerate_ghosts):.3f} ms")
print(f"_build_list (all)      {timed(run._build_list):.3f} ms")
lo = (run.x.min(0).values - 0.25).cpu().numpy(); hi = (run.x.max(0).values + 0.25).cpu().numpy()
print(f"min/max + .cpu()       {timed(lambda: ((run.x.min(0).values - 0.25).cpu().numpy(), (run.x.max(0).values + 0.25).cpu().numpy())):.3f} ms")
print(f"build_list_device      {timed(lambda: ani.build_list_device(run.ntotal, run.nlocal, run.species.data_ptr(), run.x.data_ptr(), run.cutneigh, lo, hi, stream=run._stream)):.3f} ms")
print(f"_forces after a build  {timed(lambda: (run._build_list(), run._forces()), 5):.3f} ms (build + forces)")
