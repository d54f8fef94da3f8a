"""The reference benchmark's run protocol on the device-resident loop (examples/benchmark/in.lammps:54-72): velocities at
300 K, fix langevin 300 300 100 + fix nve, dt 0.5 fs, neighbor 2.0 bin / neigh_modify every 10 check yes, 4 x 500 + 5000
warm-up steps, then the production run (default 5000 steps, `run_steps` of the input) timed as one block.

    python tools/long_run.py [atoms] [warmup] [steps] > profiles/r04_md_long_run.json

Prints one JSON line: ns/day of the production block as LAMMPS' "Performance:" line would give it (steps / wall time), with
the number of re-neighbourings it contained, the thermostat's temperature at its end and the total-energy bookkeeping."""
import json
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
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
out_scale = 0.02   # bench.py's MD model: seeded weights scaled so that the liquid stays a liquid at 300 K
sys.stdout.flush()
json_fd = os.dup(1)
os.dup2(2, 1)
dev = torch.device("cuda:0")
path = "/tmp/long_run.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024, out_scale=out_scale))
system = hx.spatial_sort(hx.water_box(atoms, seed=12345))
inp = hx.decompose(system)
ani = ani_hip.ANI(path, 0)
run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo)
run.create_velocities(300.0)
run.warm_paths()
run.run(warmup)        # `run N` without per-step output: VerletRun.run fuses the integrator halves between steps
torch.cuda.synchronize()
b0, t0 = run.nbuilds, time.perf_counter()
run.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ke, pe = run.kinetic_energy(), run.potential_energy()
out = {"what": "examples/benchmark/in.lammps protocol on md.VerletRun, one MI355X", "atoms": atoms, "warmup_steps": warmup,
       "steps": steps, "wall_s": dt, "ms_per_step": dt / steps * 1e3, "ns_per_day": steps / dt * 0.0432,
       "list_rebuilds": run.nbuilds - b0, "rebuild_interval_steps": steps / max(run.nbuilds - b0, 1),
       "temperature_K": 2.0 * ke / (3.0 * run.nlocal - 3.0) / md.BOLTZ, "energy_finite": bool(np.isfinite(pe)),
       "model_out_scale": out_scale, "mlp_arith": 1, "error_flags": ani.debug_view().error_flags}
os.write(json_fd, (json.dumps(out) + "\n").encode())
ani.close()
