"""GPU rehearsal of the multi-rank MD loop on ONE card: two processes (one per would-be GPU), brick decomposition,
ghost exchange over torch.distributed.  Backend gloo with host-staged messages, because RCCL will not put two ranks
on one device; on a multi-GPU node bench.py runs the same code with backend nccl.  The two-rank trajectory must
follow the single-rank one (same start velocities by atom tag)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NATOMS, STEPS, DT, VSIGMA = 1536, 80, 0.25, 0.03  # fastest atom ~0.11 A/fs: a re-neighbouring (with migration) every ~40 steps


def _run(rank, world, port, model_path, out_dir, overlap=False):
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
    sysm = hx.spatial_sort(hx.water_box(NATOMS))
    inp = hx.decompose(sysm, comm.grid_for(world), rank)
    ani = ani_hip.ANI(model_path, 0)
    run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, torch.device("cuda:0"), dt=DT, box_lo=sysm.boxlo, overlap=overlap)
    assert run._overlap == bool(overlap)
    # the same start velocities whatever the decomposition: a table indexed by global atom tag
    table = np.random.default_rng(99).normal(0.0, VSIGMA, size=(sysm.natoms, 3))
    run.v = torch.as_tensor(table[run.tag.cpu().numpy()], dtype=torch.float64, device="cuda:0")
    etot = [run.potential_energy() + run.kinetic_energy()]
    for _ in range(STEPS):
        run.step()
        etot.append(run.potential_energy() + run.kinetic_energy())
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), etot=np.array(etot), tag=run.tag.cpu().numpy(),
             x=run.x[: run.nlocal].cpu().numpy(), builds=run.nbuilds, nlocal0=inp.nlocal)
    ani.close()
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["plain", "overlapped-exchange"])
def test_two_rank_md_follows_single_rank(overlap, tmp_path):
    """overlapped-exchange: the two ghost exchanges of a step run on a second stream beside the rows that have no ghost
    among their candidates (the library's three-call step, include/ani_hip.h ani_step_*)."""
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import model_file as mf
    path = str(tmp_path / "gentle.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_run, args=(1, port, path, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(2, port, path, str(tmp_path), overlap), nprocs=2, join=True)
    one = np.load(tmp_path / "w1_r0.npz")
    two = [np.load(tmp_path / f"w2_r{r}.npz") for r in range(2)]
    assert int(two[0]["builds"]) >= 3      # set-up + at least two re-neighbourings with atom migration
    assert sorted(np.concatenate([d["tag"] for d in two])) == list(range(NATOMS))   # every atom has exactly one owner
    # total energy (all-reduced over ranks) step by step; fp32 forces, different summation orders
    assert np.abs(two[0]["etot"] - one["etot"]).max() < 5e-3
    assert np.array_equal(two[0]["etot"], two[1]["etot"])
    x1 = np.zeros((NATOMS, 3))
    x1[one["tag"]] = one["x"]
    x2 = np.zeros((NATOMS, 3))
    for d in two:
        x2[d["tag"]] = d["x"]
    assert np.abs(x2 - x1).max() < 1e-4   # both runs wrap owned atoms into the box at every re-neighbouring
