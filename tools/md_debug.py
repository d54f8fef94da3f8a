"""Development probe: forces of the MD loop's plain and overlapped paths against the host-pointer entry point."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
_pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, md, model_file as mf
dev = torch.device("cuda:0")
path = "/tmp/md_debug.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
sysm = hx.spatial_sort(hx.water_box(1536))
inp = hx.decompose(sysm)
for overlap in (False, True):
    ani = ani_hip.ANI(path, 0)
    run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=0.25, box_lo=sysm.boxlo, overlap=overlap)
    run.create_velocities(300.0)
    for k in range(6):
        f0 = run.f.clone()
        run.step()
        torch.cuda.synchronize()
        # reference: the same positions through a fresh handle and the host entry point
        nl = run.nlocal
        print(overlap, k, "E", run.potential_energy(), "KE", run.kinetic_energy(), "|f|max", float(run.f[:nl].abs().max()),
              "sumF", run.f[:nl].sum(0).cpu().numpy(), "ghost f max", float(run.f[nl:].abs().max()))
    ani.close()
