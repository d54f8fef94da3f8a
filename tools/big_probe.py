"""Development probe: a 10^6-atom water box on one GPU (8 periodic copies of a 124 998-atom box), list built on the device.
Extensivity is the check that needs no oracle: E(tiled) = 8 E(base), forces repeat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
import numpy as np, torch
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf, md

nbase = int(sys.argv[1]) if len(sys.argv) > 1 else 124998
t0 = time.time()
base = hx.spatial_sort(hx.water_box(nbase, seed=7))
print(f"base box {nbase} atoms generated in {time.time() - t0:.0f} s", flush=True)
L = base.boxhi - base.boxlo
path = "/tmp/big.anim"
mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=2024))
dev = torch.device("cuda", 0)


def bare_input(x, types):
    n = len(x)
    z = np.zeros(0, np.int32)
    return hx.RankInput(nlocal=n, nghost=0, x=x, types=types, tag=np.arange(n, dtype=np.int64), owner_rank=z, owner_lidx=z,
                        shift=np.zeros((0, 3), np.int32), ilist=z, numneigh=z, jlist=z, half=False)


def run(x, types, lo, box_len, steps):
    ani = ani_hip.ANI(path, 0)
    r = md.VerletRun(ani, bare_input(x, types), box_len, dev, dt=0.5, box_lo=lo)
    torch.cuda.synchronize()
    e = r.potential_energy()
    f = r.f[: r.nlocal].cpu().numpy().copy()
    t0 = time.perf_counter()
    for _ in range(steps):
        r._forces()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(steps, 1) * 1e3
    out = (e, f, ms, r.npairs, r.ntotal)
    ani.close()
    return out


e1, f1, ms1, np1, nt1 = run(base.x, base.types, base.boxlo, L, 20)
print(f"base : E = {e1:.3f} kcal/mol, {ms1:.3f} ms/force evaluation, npairs {np1}, ntotal {nt1}", flush=True)
shifts = np.array([(a, b, c) for a in range(2) for b in range(2) for c in range(2)], dtype=np.float64) * L
x8 = np.concatenate([base.x + s for s in shifts])
t8 = np.tile(base.types, 8)
e8, f8, ms8, np8, nt8 = run(x8, t8, base.boxlo, 2 * L, 10)
print(f"tiled: {len(x8)} atoms, E = {e8:.3f} kcal/mol, {ms8:.3f} ms/force evaluation = {0.0432e3 / ms8:.2f} ns/day, npairs {np8}, ntotal {nt8}")
print(f"E(tiled) / E(base) = {e8 / e1:.9f} (8 exactly);  max |f(tiled) - f(base repeated)| = {np.abs(f8 - np.tile(f1, (8, 1))).max():.3e} kcal/mol/A")
