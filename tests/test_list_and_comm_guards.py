"""Guards around the two conveniences the whole-step entry points take for granted (GPU), and the bootstrap of the native
communicator (CPU, gloo):

  * a caller's list that is NOT symmetric between owned atoms must cost the symmetric radial collection, not the forces
    (option aev_symmetric_radial, include/ani_hip.h);
  * with a communicator attached (`rcclcomm`, src/pair_ani.cpp:197-201 replaced by ani_comm_reverse) a step posts exactly
    ONE reverse exchange whatever the radial capacity does: a retry on one rank would pair its second message with the
    peers' next step;
  * NativeComm.from_torch runs the same collectives on every rank whether or not rank 0 could make an RCCL id.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import _pkg  # noqa: E402  (the spawned workers import this module without conftest.py)

_pkg.load()
from lammps_ani_amd import harness as hx  # noqa: E402

F_TOL = 2.3e-3  # kcal/mol/A = 1e-4 eV/A (north-star bar)


@pytest.fixture(scope="module")
def hip():
    from lammps_ani_amd import ani_hip
    return ani_hip


def _drop_one_direction(inp, npick=12, seed=5):
    """Remove the entry i -> j (keeping j -> i) for a few pairs of OWNED atoms that lie inside the radial cutoff."""
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(inp.numneigh)])
    keep = np.ones(inp.jlist.shape[0], dtype=bool)
    picked = 0
    for i in rng.permutation(inp.nlocal):
        seg = inp.jlist[off[i]:off[i + 1]]
        d = np.linalg.norm(inp.x[seg] - inp.x[i], axis=1)
        cand = np.nonzero((seg < inp.nlocal) & (d < 4.5))[0]
        if cand.size == 0:
            continue
        keep[off[i] + cand[0]] = False
        picked += 1
        if picked == npick:
            break
    assert picked == npick
    num = inp.numneigh.copy()
    drop_i = np.repeat(np.arange(inp.nlocal), inp.numneigh)[~keep]
    np.subtract.at(num, drop_i, 1)
    return hx.RankInput(nlocal=inp.nlocal, nghost=inp.nghost, x=inp.x, types=inp.types, tag=inp.tag, owner_rank=inp.owner_rank,
                        owner_lidx=inp.owner_lidx, shift=inp.shift, ilist=inp.ilist, numneigh=num.astype(np.int32),
                        jlist=np.ascontiguousarray(inp.jlist[keep]), half=False)


@pytest.mark.gpu
def test_one_sided_list_is_detected_and_still_matches_the_oracle(model_cache, hip, capfd):
    from oracle import Oracle
    p = model_cache("ani2x", 2, 2024)
    inp = hx.decompose(hx.water_box(384, seed=11))
    ani = hip.ANI(p, 0)
    sym = ani.compute(inp, ago=0)
    assert ani.debug_view().error_flags & 8 == 0          # a LAMMPS-style full list passes the check
    lop = _drop_one_direction(inp)
    got = ani.compute(lop, ago=0)
    assert ani.debug_view().error_flags & 8               # ... a one-sided one is noticed,
    assert "not symmetric" in capfd.readouterr().err      # said once,
    ref = Oracle(p).compute(lop)                           # and evaluated as it stands: E_i sees r_ij, E_j does not
    assert np.abs(got["force"] - ref["force"]).max() < F_TOL
    assert abs(got["energy"] - ref["energy"]) < 2e-3
    assert np.abs(got["force"] - sym["force"]).max() > 1e-2   # the dropped entries mattered
    again = ani.compute(inp, ago=0)                        # the next symmetric epoch collects again, same answer as before
    assert np.abs(again["force"] - sym["force"]).max() < 1e-4
    ani.close()


def _sphere_in_a_box(n=140, L=12.5):
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    th = np.pi * (1 + 5 ** 0.5) * k
    pts = 2.5 * np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1) + L / 2
    return hx.System(pts, np.full(n, 1, np.int32), np.zeros(3), np.full(3, L), (True,) * 3)


@pytest.mark.gpu
def test_capacity_overflow_with_a_communicator_posts_one_exchange_per_step(model_cache, hip, capfd):
    """Atoms on a 2.5 A sphere in a periodic box: every owned pair is inside Rcr, so the screened radial capacity (3/4 of the
    list, at least 128) overflows and the plain host entry point repeats the step; the images make ghosts, so a communicator
    has something to move.  With one attached: no retry, one reverse exchange per call, same forces."""
    import torch
    p = model_cache("ani2x", 1, 2024)
    inp = hx.decompose(_sphere_in_a_box())
    assert inp.nghost > 0 and inp.numneigh.max() >= 139
    plain = hip.ANI(p, 0)
    ref = plain.compute(inp, ago=0)
    assert "full_radial_capacity = 1" in capfd.readouterr().err     # the geometry does overflow the 3/4 estimate
    plain.close()
    folded = ref["force"][: inp.nlocal].copy()
    np.add.at(folded, inp.owner_lidx, ref["force"][inp.nlocal:])

    nat = hip.NativeComm(1, 0, hip.NativeComm.unique_id(), 0)
    nat.set_option("self_through_rccl", 1)                          # the own chunk goes through ncclSend / ncclRecv
    dev = torch.device("cuda:0")
    idx = torch.as_tensor(inp.owner_lidx.astype(np.int64), device=dev)
    shift = torch.zeros((inp.nghost, 3), dtype=torch.float64, device=dev)
    nat.set_epoch([inp.nghost], [inp.nghost], idx, shift)
    ani = hip.ANI(p, 0)
    ani.attach_comm(nat)
    got = ani.compute(inp, ago=0)
    err = capfd.readouterr().err
    assert "continuing with full_radial_capacity" not in err         # no second run_step + exchange
    assert nat.stat("reverse_exchanges") == 1
    assert np.abs(got["force"][: inp.nlocal] - folded).max() < 5e-2 and abs(got["energy"] - ref["energy"]) < 5e-2
    assert np.abs(got["force"][inp.nlocal:]).max() == 0.0            # ghost rows were folded on the device
    ani.compute(inp, ago=1)
    assert nat.stat("reverse_exchanges") == 2
    assert nat.stat("broken") == 0
    ani.attach_comm(None)
    ani.close()

    # attached in the middle of an epoch (list cached with the 3/4 capacity): the lists are re-sized before the step, once
    inp2 = hx.decompose(hx.water_box(192, seed=3))
    idx2 = torch.as_tensor(inp2.owner_lidx.astype(np.int64), device=dev)
    shift2 = torch.zeros((inp2.nghost, 3), dtype=torch.float64, device=dev)
    ani = hip.ANI(p, 0)
    a = ani.compute(inp2, ago=0)
    fa = a["force"][: inp2.nlocal].copy()
    np.add.at(fa, inp2.owner_lidx, a["force"][inp2.nlocal:])
    nat.set_epoch([inp2.nghost], [inp2.nghost], idx2, shift2)
    ani.attach_comm(nat)
    b = ani.compute(inp2, ago=1)
    assert nat.stat("reverse_exchanges") == 3
    assert np.abs(b["force"][: inp2.nlocal] - fa).max() < 1e-3
    ani.attach_comm(None)
    ani.close()
    nat.close()


# ---- bootstrap of the native communicator: the same collectives on every rank, whatever fails where (CPU, gloo) ----------
def _boot_worker(rank, world, port, out_dir, fail_on_rank0):
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    import torch.distributed as dist
    from lammps_ani_amd import ani_hip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if fail_on_rank0:
        def boom():
            raise ani_hip.AniError("librccl.so.1 not found (simulated)")
        ani_hip.NativeComm.unique_id = staticmethod(boom)
    else:
        # an id exists, creating the communicator fails everywhere (no HIP device in this container / not two per card):
        # what matters is that every rank comes back with the same verdict instead of waiting for the others
        ani_hip.NativeComm.unique_id = staticmethod(lambda: bytes(128))
    try:
        ani_hip.NativeComm.from_torch(0)
        verdict = "made"
    except ani_hip.AniError as e:
        verdict = "AniError: " + str(e)
    # the fallback a caller takes next (bench.py): a collective all ranks must reach together
    import torch
    t = torch.ones(1)
    dist.all_reduce(t)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{verdict}\n{float(t)}\n")
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_on_rank0", [True, False])
def test_native_comm_bootstrap_fails_on_all_ranks_or_none(tmp_path, fail_on_rank0):
    import torch
    import torch.multiprocessing as mp
    if not fail_on_rank0 and torch.cuda.device_count() > 0:
        pytest.skip("with a HIP device present ani_comm_create would really start ncclCommInitRank on a made-up id")
    world = 2
    port = 29500 + (os.getpid() % 2000) + 131 + int(fail_on_rank0)
    mp.spawn(_boot_worker, args=(world, port, str(tmp_path), fail_on_rank0), nprocs=world, join=True)
    lines = [open(tmp_path / f"r{r}.txt").read().splitlines() for r in range(world)]
    for v, t in lines:
        assert v.startswith("AniError"), v
        assert float(t) == float(world)       # every rank reached the fallback's collective
    if fail_on_rank0:
        assert all("rank 0 could not make an RCCL id" in v and "simulated" in v for v, _ in lines)
    else:
        assert all("ani_comm_create failed" in v for v, _ in lines)
