"""The message layout of the device exchange (include/ani_comm.h) against the torch.distributed path, on CPU.

`ani_comm_forward` / `ani_comm_reverse` move, for every peer, one chunk of a packed buffer: chunk offsets come from
`ani_comm_plan` (host arithmetic inside libani_hip.so, no GPU needed).  Here the same maps `comm.DomainComm` builds
(send_idx, send_shift, per-peer counts) are pushed through point-to-point gloo messages laid out by that plan — what the
grouped ncclSend / ncclRecv do on the GPU box — and must give what DomainComm's own all_to_all_single path gives:
ghost positions after a forward exchange, owner sums after a reverse exchange.  World sizes 2 and 4.
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


def _p2p_alltoall(send, so, sc, recv, ro, rc, rank, world):
    """chunk p of `send` to rank p, chunk p of `recv` from rank p: the grouped send/recv of ani_comm.cpp's a2a_bytes"""
    reqs = []
    for p in range(world):
        if p == rank:
            recv[ro[p]: ro[p] + rc[p]] = send[so[p]: so[p] + sc[p]]
            continue
        if sc[p] > 0:
            reqs.append(dist.isend(send[so[p]: so[p] + sc[p]].contiguous(), dst=p))
        if rc[p] > 0:
            reqs.append(dist.irecv(recv[ro[p]: ro[p] + rc[p]], src=p))
    for r in reqs:
        r.wait()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import ani_hip, comm, harness as hx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = hx.random_box(400, 3, 19.0, seed=33, min_dist=1.0)
    grid = comm.grid_for(world)
    inp = hx.decompose(s, grid, rank, cutoff=5.1, skin=2.0)
    dc = comm.DomainComm(grid, s.boxlo, s.boxhi - s.boxlo, CUT, torch.device("cpu"))
    n = inp.nlocal
    x = torch.from_numpy(inp.x[:n].copy())
    tag = torch.from_numpy(inp.tag[:n].astype(np.int64))
    xo, tago = dc.exchange(x, tag)
    xa, taga = dc.borders(xo, tago)
    nl, ng = dc.nlocal, dc.nghost
    so, ro, nsend, nrecv = ani_hip.comm_plan(dc.send_splits, dc.recv_splits)
    assert nsend == dc.send_idx.numel() and nrecv == ng
    sc, rc = dc.send_splits, dc.recv_splits
    # forward: owners move, ghosts follow -- the torch path against the planned point-to-point messages
    xm = xa.clone()
    xm[:nl] += torch.from_numpy(np.random.default_rng(100 + rank).normal(0.0, 0.05, size=(nl, 3)))
    x_ref = xm.clone()
    dc.forward_positions(x_ref)
    x_p2p = xm.clone()
    packed = x_p2p[:nl][dc.send_idx] + dc.send_shift           # ani_md_pack_ghosts
    ghosts = torch.empty((ng, 3), dtype=torch.float64)
    _p2p_alltoall(packed, so, sc, ghosts, ro, rc, rank, world)
    x_p2p[nl:] = ghosts
    # reverse: the roles of the counts swap (ani_comm_reverse_send), then unpack adds into send_idx
    f = torch.from_numpy(np.random.default_rng(200 + rank).normal(size=(nl + ng, 3)))
    f_ref = f.clone()
    dc.reverse_add(f_ref)
    f_p2p = f.clone()
    staged = torch.empty((nsend, 3), dtype=torch.float64)
    _p2p_alltoall(f_p2p[nl:].contiguous(), ro, rc, staged, so, sc, rank, world)
    f_p2p[:nl].index_add_(0, dc.send_idx, staged)             # ani_md_unpack_reverse
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), dx=(x_ref - x_p2p).abs().max().item(),
             df=(f_ref[:nl] - f_p2p[:nl]).abs().max().item(), ng=ng, nsend=nsend)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_planned_point_to_point_exchange_equals_the_all_to_all_path(world, tmp_path):
    port = 29500 + (os.getpid() % 2000) + 60 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        d = np.load(tmp_path / f"r{rank}.npz")
        assert int(d["ng"]) > 0 and int(d["nsend"]) > 0
        assert float(d["dx"]) == 0.0
        assert float(d["df"]) < 1e-12   # same addends, the order of the owner sums may differ


def test_plan_layout_and_argument_checks():
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from lammps_ani_amd import ani_hip
    so, ro, ns, nr = ani_hip.comm_plan([3, 0, 5, 2], [1, 2, 0, 4])
    assert so.tolist() == [0, 3, 3, 8] and ro.tolist() == [0, 1, 3, 3] and (ns, nr) == (10, 7)
    with pytest.raises(ani_hip.AniError):
        ani_hip.comm_plan([1, -1], [0, 0])
    lib = ani_hip.lib()
    for name in ani_hip.COMM_EXPORTS:   # the C ABI of include/ani_comm.h is all there
        assert hasattr(lib, name), name
