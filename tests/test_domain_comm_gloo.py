"""World-size 1, 2 and 4 gloo tests of comm.DomainComm — the device-side stand-in for LAMMPS' Comm::exchange /
Comm::borders / forward_comm / reverse_comm that the multi-GPU MD loop runs between force evaluations.  The oracle is
the host harness (lammps_ani_amd.harness.decompose, the LAMMPS stand-in every other test uses): after a random
displacement that carries atoms across brick faces and box faces,

  * exchange()  leaves every rank with exactly the atoms the harness assigns to its brick (by global tag);
  * borders()   produces exactly the harness's ghost shell (as a multiset of (tag, image position));
  * forward_positions() keeps ghosts equal to owner + image shift after a further displacement;
  * reverse_add() returns to every owned atom the sum over its ghost copies on all ranks.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUT = 7.1


def _system():
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import harness as hx
    return hx, hx.random_box(400, 3, 19.0, seed=33, min_dist=1.0)


def _worker(rank, world, port, out_dir):
    hx, s = _system()
    from lammps_ani_amd import comm
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = comm.grid_for(world)
    inp = hx.decompose(s, grid, rank, cutoff=5.1, skin=2.0)
    L = s.boxhi - s.boxlo
    dc = comm.DomainComm(grid, s.boxlo, L, CUT, torch.device("cpu"))
    # owned atoms of the start decomposition, displaced (same displacement table on every rank, indexed by tag)
    disp = np.random.default_rng(7).normal(0.0, 1.5, size=(s.natoms, 3))
    n = inp.nlocal
    tag = torch.from_numpy(inp.tag[:n].astype(np.int64))
    x = torch.from_numpy(inp.x[:n] + disp[inp.tag[:n]])
    v = torch.from_numpy(disp[inp.tag[:n]].copy())          # any per-atom payload: must arrive with its atom
    xo, vo, tago = dc.exchange(x, v, tag)
    xa, taga = dc.borders(xo, tago)
    # forward: move owners again, refresh ghosts
    disp2 = np.random.default_rng(8).normal(0.0, 0.05, size=(s.natoms, 3))
    xm = xa.clone()
    xm[: dc.nlocal] += torch.from_numpy(disp2[tago.numpy()])
    dc.forward_positions(xm)
    ghost_err = (xm[dc.nlocal:] - (xa[dc.nlocal:] + torch.from_numpy(disp2[taga[dc.nlocal:].numpy()]))).abs().max().item() if dc.nghost else 0.0
    # reverse: every ghost carries (1, tag, 0); owners must receive (number of ghost copies, copies * tag, 0)
    f = torch.zeros((dc.nlocal + dc.nghost, 3), dtype=torch.float64)
    f[dc.nlocal:, 0] = 1.0
    f[dc.nlocal:, 1] = taga[dc.nlocal:].double()
    dc.reverse_add(f)
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), tag=tago.numpy(), x=xo.numpy(), v=vo.numpy(),
             gtag=taga[dc.nlocal:].numpy(), gx=xa[dc.nlocal:].numpy(), ghost_err=ghost_err, rev=f[: dc.nlocal].numpy())
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])   # 8 = the 2 x 2 x 2 bricks of the 8-GPU node: every rank borders every other
def test_domain_comm_matches_harness(world, tmp_path):
    hx, s = _system()
    from lammps_ani_amd import comm
    port = 29500 + (os.getpid() % 2000) + 20 + world
    if world == 1:
        _worker(0, 1, port, str(tmp_path))
    else:
        mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    disp = np.random.default_rng(7).normal(0.0, 1.5, size=(s.natoms, 3))
    moved = hx.System(s.x + disp, s.types, s.boxlo, s.boxhi, s.periodic)
    grid = comm.grid_for(world)
    copies = np.zeros(s.natoms)
    seen = 0
    for rank in range(world):
        ref = hx.decompose(moved, grid, rank, cutoff=5.1, skin=2.0)
        d = np.load(tmp_path / f"w{world}_r{rank}.npz")
        nl = ref.nlocal
        # owned atoms: same set, positions wrapped like the harness wraps them, payload intact
        assert sorted(d["tag"]) == sorted(ref.tag[:nl])
        order_ref = np.argsort(ref.tag[:nl]); order_got = np.argsort(d["tag"])
        np.testing.assert_allclose(d["x"][order_got], ref.x[:nl][order_ref], atol=1e-9)
        np.testing.assert_allclose(d["v"][order_got], disp[np.sort(d["tag"])], atol=0)
        # ghosts: the same multiset of (tag, image position)
        key = lambda t, x: sorted((int(a), round(float(b[0]), 7), round(float(b[1]), 7), round(float(b[2]), 7)) for a, b in zip(t, x))
        assert key(d["gtag"], d["gx"]) == key(ref.tag[nl:], ref.x[nl:])
        assert float(d["ghost_err"]) < 1e-12
        np.add.at(copies, ref.tag[nl:], 1.0)
        seen += nl
    assert seen == s.natoms
    for rank in range(world):
        d = np.load(tmp_path / f"w{world}_r{rank}.npz")
        np.testing.assert_allclose(d["rev"][:, 0], copies[d["tag"]], atol=0)
        np.testing.assert_allclose(d["rev"][:, 1], copies[d["tag"]] * d["tag"], atol=0)
        assert np.all(d["rev"][:, 2] == 0)
