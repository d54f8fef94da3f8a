"""Scratch probe: NVE energy conservation and list-rebuild behaviour of lammps_ani_amd.md.VerletRun on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _pkg; _pkg.load()
from lammps_ani_amd import ani_hip, harness, model_file, md

natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
single = int(sys.argv[4]) if len(sys.argv) > 4 else 1
T0 = float(sys.argv[5]) if len(sys.argv) > 5 else 300.0
margin = 0.0
lang = (300.0, 100.0) if os.environ.get('LANGEVIN') else None
path = "/tmp/md_probe.anim"
model_file.write_model(path, model_file.synthetic_model("ani2x", 1, seed=1, out_scale=float(os.environ.get("OUT_SCALE", "0.02"))))
sysm = harness.spatial_sort(harness.water_box(natoms))
inp = harness.decompose(sysm, skin=2.0 + margin)
ani = ani_hip.ANI(path, 0, use_single=bool(single))
if os.environ.get("MLP_ARITH"):   # 2 (default): two-term fp16 splits, 1: exact bf16 splits, 0: fp32-input MFMA
    ani.set_option("mlp_arith", int(os.environ["MLP_ARITH"]))
    print("mlp_arith", os.environ["MLP_ARITH"])
dev = torch.device("cuda:0")
run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=dt, ghost_margin=margin, box_lo=sysm.boxlo, langevin=lang)
# list check against the harness at set-up
inp7 = harness.decompose(sysm, skin=2.0 + margin)  # ghosts identical; host list has the wider cutoff, so rebuild at 7.1
nn, jl = ani.debug_list(inp.nlocal)
print("npairs", run.npairs, "mean numneigh", nn.mean(), "max", nn.max())
run.create_velocities(T0)
e0 = run.potential_energy() + run.kinetic_energy()
print(f"step 0 pe {run.potential_energy():.4f} ke {run.kinetic_energy():.4f} etot {e0:.4f}")
torch.cuda.synchronize(); t0 = time.time()
for s in range(1, steps + 1):
    run.step()
    if s % max(1, steps // 10) == 0:
        pe, ke = run.potential_energy(), run.kinetic_energy()
        print(f"step {s} pe {pe:.4f} ke {ke:.4f} etot {pe+ke:.4f} drift {pe+ke-e0:+.5f} T {2*ke/(3*run.nlocal-3)/md.BOLTZ:.1f} builds {run.nbuilds}")
torch.cuda.synchronize(); print("ms/step", (time.time() - t0) / steps * 1e3)
