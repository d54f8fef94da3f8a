"""Development probe: where a re-neighbouring step of md.VerletRun spends its time (100 002-atom water box, one rank)."""
import sys, time
sys.path.insert(0, ".")
import _pkg; _pkg.load()
import numpy as np, torch
from lammps_ani_amd import ani_hip, harness as hx, md, model_file as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
dev = torch.device("cuda:0")
path = "/tmp/probe.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024, out_scale=0.02))
system = hx.spatial_sort(hx.water_box(n, seed=12345))
inp = hx.decompose(system)
ani = ani_hip.ANI(path, 0)
run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo)
run.create_velocities(300.0)
for _ in range(20):
    run.step()

def timed(fn, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

nl = run.nlocal
print(f"whole _build_list        {timed(run._build_list):.3f} ms")
print(f"  dc.exchange            {timed(lambda: run.dc.exchange(run.x[:nl], run.v, run.tag, run.species[:nl])):.3f} ms")
xo, v, tag, sp = run.dc.exchange(run.x[:nl], run.v, run.tag, run.species[:nl])
print(f"  dc.borders             {timed(lambda: run.dc.borders(xo, sp.contiguous())):.3f} ms")
print(f"  per-atom factors       {timed(run._per_atom_factors):.3f} ms")
lo = run.dc.sub_lo - run.cutneigh - 0.25; hi = run.dc.sub_hi + run.cutneigh + 0.25
print(f"  build_list_device      {timed(lambda: ani.build_list_device(run.ntotal, nl, run.species.data_ptr(), run.x.data_ptr(), run.cutneigh, lo, hi, stream=run._stream)):.3f} ms")
print(f"step without rebuild     {timed(run.step, 9):.3f} ms")
