"""Trajectory-level check in the shape of the reference's tests/test_lmp_with_ase.py:114-157,167-205 (GPU).

The reference runs 4 NVE steps of 0.1 fs on the 30-atom water box under LAMMPS (`neigh_modify every 2 delay 0 check no`, zero
start velocities, tests/in.lammps) and under ASE's VelocityVerlet with the torchani calculator, and compares per step:
forces, positions, temperature, potential energy and (pyaev) the stress with its kinetic part, on 1 and 2 MPI ranks, at
fp32 `atol = rtol = 1e-3` / fp64 `atol 1e-9, rtol 1e-5`, temperature `atol 0.13`.  Here the LAMMPS side is the device-resident
loop (md.VerletRun: device neighbour list, ghost exchange, libani_hip through the C ABI) and the ASE side is a plain numpy
velocity Verlet whose forces, energy and virial come from oracle/ (test infrastructure) on a host-built LAMMPS-style list —
same matrix: {30-atom PBC box of the reference, 1 500-atom box} x {single, double} x {1, 2 ranks (gloo, two processes on the
card)}.  Until trained weights exist this also keeps the yaml `run_*` runner (tests/test_reference_yaml.py) honest: it is the
same loop.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load()
from lammps_ani_amd import harness as hx  # noqa: E402

pytestmark = pytest.mark.gpu

STEPS, DT = 4, 0.1                      # tests/test_lmp_with_ase.py:22, tests/in.lammps "timestep 0.1"
FTM2V = 1.0 / 48.88821291 / 48.88821291  # LAMMPS units real
MVV2E = 48.88821291 * 48.88821291
BOLTZ = 0.0019872067
MASSES = np.array([1.008, 12.011, 14.007, 15.999, 32.06, 18.998, 35.45])


def _system(name):
    if name == "water30":
        return hx.read_lammps_data(os.path.join(ROOT, "tests", "golden", "water-0.8nm.data"))   # the reference's own box
    return hx.spatial_sort(hx.water_box(1500, seed=4))


def _oracle_trajectory(sysm, model_path):
    """numpy velocity Verlet (LAMMPS fix nve order) with the oracle as the calculator; the list is rebuilt every step (the
    reference's ASE side has no list to age)."""
    from oracle import Oracle
    o = Oracle(model_path)
    n = sysm.natoms
    m = MASSES[sysm.types - 1][:, None]
    x, v = sysm.x.copy(), np.zeros((n, 3))

    def evaluate(xx):
        inp = hx.decompose(sysm, x_override=xx)
        r = o.compute(inp)
        f = np.zeros((n, 3))
        np.add.at(f, inp.tag[: inp.nlocal], r["force"][: inp.nlocal])
        np.add.at(f, inp.tag[inp.nlocal:], r["force"][inp.nlocal:])       # ghosts carry their owner's tag
        return f, r["energy"], r["virial"]

    def record(f, e, w):
        ke = 0.5 * MVV2E * float((m * v * v).sum())
        kin = MVV2E * np.einsum("i,ia,ib->ab", m[:, 0], v, v)
        return dict(f=f.copy(), x=x.copy(), T=2.0 * ke / ((3.0 * n - 3.0) * BOLTZ), pe=e, stress_v=kin + w)

    f, e, w = evaluate(x)
    out = [record(f, e, w)]
    for _ in range(STEPS):
        v += 0.5 * DT * FTM2V * f / m
        x += DT * v
        f, e, w = evaluate(x)
        v += 0.5 * DT * FTM2V * f / m
        out.append(record(f, e, w))
    return out


def _hip_worker(rank, world, port, model_path, sysname, single, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    import torch
    import torch.distributed as dist
    from lammps_ani_amd import ani_hip, comm, md, harness as hx
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sysm = _system(sysname)
    n = sysm.natoms
    inp = hx.decompose(sysm, comm.grid_for(world), rank)
    dev = torch.device("cuda:0")
    ani = ani_hip.ANI(model_path, 0, -1, use_single=single)
    run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=DT, every=2, box_lo=sysm.boxlo, vflag=True)
    masses = torch.as_tensor(MASSES, device=dev)
    rec = {}

    def record(k):
        tag = run.tag.cpu().numpy()
        mm = masses[run.species[: run.nlocal].long()]
        kin = MVV2E * torch.einsum("i,ia,ib->ab", mm, run.v, run.v).cpu().numpy()
        rec[f"tag{k}"], rec[f"f{k}"], rec[f"x{k}"] = tag, run.f[: run.nlocal].cpu().numpy(), run.x[: run.nlocal].cpu().numpy()
        rec[f"kin{k}"] = kin
        rec[f"T{k}"], rec[f"pe{k}"], rec[f"w{k}"] = run.temperature(n), run.potential_energy(), run.virial()

    record(0)
    for k in range(1, STEPS + 1):
        run.step(force_rebuild=(k % 2 == 0))       # neigh_modify every 2 delay 0 check no
        record(k)
    rec["builds"] = run.nbuilds
    np.savez(os.path.join(out_dir, f"hip_w{world}_r{rank}.npz"), **rec)
    ani.close()
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2], ids=["num_tasks_1", "num_tasks_2"])
@pytest.mark.parametrize("single", [True, False], ids=["precision_single", "precision_double"])
@pytest.mark.parametrize("sysname", ["water30", "water1500"])
def test_four_nve_steps_follow_the_oracle_driven_integrator(sysname, single, world, tmp_path, model_cache):
    import torch.multiprocessing as mp
    model_path = model_cache("ani2x", 8, 2024)          # full ensemble, as `ani_num_models -1` in the reference's matrix
    sysm = _system(sysname)
    n, L = sysm.natoms, sysm.boxhi - sysm.boxlo
    ref = _oracle_trajectory(sysm, model_path)
    port = 29500 + (os.getpid() % 2000) + 211 + 2 * world + int(single)
    mp.spawn(_hip_worker, args=(world, port, model_path, sysname, single, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"hip_w{world}_r{r}.npz") for r in range(world)]
    assert int(ranks[0]["builds"]) == 3                 # set-up + steps 2 and 4
    atol, rtol = (1e-3, 1e-3) if single else (1e-9, 1e-5)
    for k in range(STEPS + 1):
        f, x, kin = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((3, 3))
        owners = np.zeros(n, dtype=int)
        for d in ranks:
            f[d[f"tag{k}"]], x[d[f"tag{k}"]] = d[f"f{k}"], d[f"x{k}"]
            owners[d[f"tag{k}"]] += 1
            kin += d[f"kin{k}"]
        assert (owners == 1).all()
        r = ref[k]
        dx = x - r["x"]
        dx -= L * np.round(dx / L)                      # the loop wraps owned atoms into the box at a re-neighbouring
        d0 = ranks[0]
        print(f"{sysname} step {k}: max |dF| {np.abs(f - r['f']).max():.2e}  |dx| {np.abs(dx).max():.2e}  dT {abs(float(d0[f'T{k}']) - r['T']):.2e}  "
              f"dPE {abs(float(d0[f'pe{k}']) - r['pe']):.2e}  d(stress V) {np.abs(kin + d0[f'w{k}'] - r['stress_v']).max():.2e}")
        assert np.allclose(f, r["f"], rtol, atol)                                   # compare force
        assert np.allclose(dx, 0.0, rtol, max(atol, 1e-12))                         # compare position
        assert np.allclose(float(d0[f"T{k}"]), r["T"], atol=1.3e-1)                # compare temperature
        pe_atol = atol if single else 1e-9 * max(1.0, abs(r["pe"]) * 1e-3)          # fp64 sums of 5e5 kcal/mol: 9e-9 relative in the yamls
        assert np.allclose(float(d0[f"pe{k}"]), r["pe"], rtol, pe_atol)             # compare potential energy
        assert np.allclose(kin + d0[f"w{k}"], r["stress_v"], rtol, max(atol, 1e-7))  # stress x volume, kinetic part included
    # the trajectory moved: forces changed between the first and the last step by far more than the tolerance
    assert np.abs(ref[STEPS]["f"] - ref[0]["f"]).max() > 10 * atol
