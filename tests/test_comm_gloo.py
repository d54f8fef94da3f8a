"""World-size-2 (and 4) gloo tests of the multi-rank path: brick decomposition + ghost-force reverse exchange +
ghost-position forward exchange.  The per-rank force evaluation is the CPU oracle (allowed in tests); on the GPU box
bench.py runs the same GhostExchange with libani_hip and backend nccl (RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, model_path, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import comm, harness as hx
    from oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = hx.random_box(90, 3, 18.0, seed=21, min_dist=1.1)
    grid = comm.grid_for(world)
    inp = hx.decompose(s, grid, rank, cutoff=5.1, skin=2.0)
    ex = comm.GhostExchange(inp, s.boxhi - s.boxlo, torch.device("cpu"))
    o = Oracle(model_path)
    r = o.compute(inp)
    f = torch.from_numpy(r["force"].copy())
    ex.reverse_add(f)
    e = torch.tensor([r["energy"]], dtype=torch.float64)
    dist.all_reduce(e)
    # forward exchange: displace local atoms, refresh ghosts, compare with a fresh decomposition of the moved system
    rng = np.random.default_rng(5)
    disp = rng.normal(0, 0.05, size=(s.natoms, 3))
    x = torch.from_numpy(inp.x.copy())
    x[: inp.nlocal] += torch.from_numpy(disp[inp.tag[: inp.nlocal]])
    ex.forward_positions(x)
    expect = inp.x.copy()
    expect += disp[inp.tag]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), tag=inp.tag[: inp.nlocal], f=f[: inp.nlocal].numpy(), e=e.numpy(),
             xerr=np.abs(x.numpy() - expect).max())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ghost_exchange_matches_single_rank(world, tmp_path, model_cache):
    sys.path.insert(0, ROOT)
    from lammps_ani_amd import harness as hx
    from oracle import Oracle
    p = model_cache("tiny", 2, 5)
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, p, str(tmp_path)), nprocs=world, join=True)
    s = hx.random_box(90, 3, 18.0, seed=21, min_dist=1.1)
    inp = hx.decompose(s)
    r = Oracle(p).compute(inp)
    f1 = r["force"][: inp.nlocal].copy()
    np.add.at(f1, inp.owner_lidx, r["force"][inp.nlocal:])
    F = np.zeros_like(f1)
    for rank in range(world):
        d = np.load(tmp_path / f"r{rank}.npz")
        F[d["tag"]] = d["f"]
        assert abs(float(d["e"][0]) - r["energy"]) < 1e-6
        assert float(d["xerr"]) < 1e-12
    np.testing.assert_allclose(F, f1, rtol=0, atol=1e-9)
