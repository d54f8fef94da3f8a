"""GPU tests of SURVEY.md §8 row f1: the device-side neighbour list (ani_build_list_device) and the device-resident
timestep loop around it (lammps_ani_amd.md.VerletRun).

  * the list is integer work: bit-exact against the harness's host build (same neighbour SETS per centre; the
    order inside a centre's segment is free);
  * forces from the device-built list equal forces from the host list to the fp32 force tolerance (only the
    summation order differs);
  * velocity-Verlet NVE through several re-neighbourings conserves E_pot + E_kin: the size-independent property that
    ties forces, energy, list rebuilds and ghost exchange together.
"""
import numpy as np
import pytest

from lammps_ani_amd import harness as hx
from lammps_ani_amd import model_file as mf

pytestmark = pytest.mark.gpu

F_TOL = 2.3e-3  # kcal/mol/A, as tests/test_hip_parity.py


@pytest.fixture(scope="module")
def hip():
    from lammps_ani_amd import ani_hip
    return ani_hip


def _segments(numneigh, jlist):
    off = np.concatenate([[0], np.cumsum(numneigh)])
    return [np.sort(jlist[off[i]:off[i + 1]]) for i in range(len(numneigh))]


def _build(ani, inp, cutneigh=7.1, pad=0.25):
    import torch
    dev = torch.device("cuda:0")
    x = torch.as_tensor(inp.x, dtype=torch.float64, device=dev).contiguous()
    sp = torch.as_tensor(inp.species.astype(np.int32), device=dev)
    lo, hi = inp.x.min(0) - pad, inp.x.max(0) + pad
    n = ani.build_list_device(inp.ntotal, inp.nlocal, sp.data_ptr(), x.data_ptr(), cutneigh, lo, hi)
    torch.cuda.synchronize()
    return n, x, sp


@pytest.mark.parametrize("case", ["water_1rank", "mixed_rank1of2", "mixed_rank5of8", "tight_box"])
def test_device_list_equals_host_list(case, model_cache, hip):
    if case == "water_1rank":
        sysm, grid, rank = hx.spatial_sort(hx.water_box(3000)), (1, 1, 1), 0
    elif case == "mixed_rank1of2":
        sysm, grid, rank = hx.random_box(1500, 7, 26.0, seed=3), (2, 1, 1), 1
    elif case == "mixed_rank5of8":
        sysm, grid, rank = hx.random_box(2500, 7, 31.0, seed=4), (2, 2, 2), 5
    else:  # box barely larger than the cutoff: one cell per dimension plus clamped ghosts
        sysm, grid, rank = hx.random_box(300, 7, 15.0, seed=5), (1, 1, 1), 0
    inp = hx.decompose(sysm, grid=grid, rank=rank)
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    # a bounding box that does NOT cover all atoms must give the same list (edge-cell clamping)
    for pad in (0.25, -3.0):
        n, _, _ = _build(ani, inp, pad=pad)
        assert n == inp.npairs
        nn, jl = ani.debug_list(inp.nlocal)
        assert np.array_equal(nn, inp.numneigh)
        for a, b in zip(_segments(nn, jl), _segments(inp.numneigh, inp.jlist)):
            assert np.array_equal(a, b)
    ani.close()


def test_device_list_is_deterministic_and_handles_empty(model_cache, hip):
    sysm = hx.random_box(800, 7, 22.0, seed=8)
    inp = hx.decompose(sysm)
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    _build(ani, inp)
    a = ani.debug_list(inp.nlocal)
    _build(ani, inp)
    b = ani.debug_list(inp.nlocal)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a rank that owns nothing
    empty = hx.RankInput(nlocal=0, nghost=inp.ntotal, x=inp.x, types=inp.types, tag=inp.tag,
                         owner_rank=np.zeros(inp.ntotal, np.int32), owner_lidx=np.zeros(inp.ntotal, np.int32),
                         shift=np.zeros((inp.ntotal, 3), np.int32), ilist=np.zeros(0, np.int32),
                         numneigh=np.zeros(0, np.int32), jlist=np.zeros(0, np.int32), half=False)
    n, _, _ = _build(ani, empty)
    assert n == 0
    ani.close()


def test_forces_from_device_list_match_host_list(model_cache, hip):
    import torch
    sysm = hx.spatial_sort(hx.water_box(3000))
    inp = hx.decompose(sysm)
    ani = hip.ANI(model_cache("ani2x", 2, 11), 0)
    ref = ani.compute(inp, ago=0)
    n, x, sp = _build(ani, inp)
    f = torch.zeros((inp.ntotal, 3), dtype=torch.float64, device=x.device)
    ev = torch.zeros(10, dtype=torch.float64, device=x.device)
    ani.compute_device(inp.ntotal, inp.nlocal, None, x.data_ptr(), n, None, None, None, 1, f.data_ptr(), ev.data_ptr(),
                       vflag=True)
    torch.cuda.synchronize()
    assert abs(float(ev[0]) - ref["energy"]) < 2e-3 * 30
    assert np.abs(f.cpu().numpy() - ref["force"]).max() < F_TOL
    assert np.abs(ev[1:].cpu().numpy().reshape(3, 3) - ref["virial"]).max() < 2e-2 * 30
    ani.close()


def test_host_array_list_build_equals_host_list(model_cache, hip):
    """ani_build_list (host species/positions in, list built on the device) followed by the host-pointer compute at
    ago != 0 -- what `pair_style ani ... devlist` does -- against the harness list through the same entry point."""
    sysm = hx.random_box(2500, 7, 31.0, seed=4)
    inp = hx.decompose(sysm, grid=(2, 2, 2), rank=5)
    ani = hip.ANI(model_cache("ani2x", 2, 11), 0)
    ref = ani.compute(inp, ago=0)
    n = ani.build_list(inp.species, inp.x, inp.nlocal, 7.1)
    assert n == inp.npairs
    nn, jl = ani.debug_list(inp.nlocal)
    assert np.array_equal(nn, inp.numneigh)
    for a, b in zip(_segments(nn, jl), _segments(inp.numneigh, inp.jlist)):
        assert np.array_equal(a, b)
    out = ani.compute(inp, ago=1)   # list pointers are ignored at ago != 0: the installed (device-built) list is used
    assert abs(out["energy"] - ref["energy"]) < 1e-3
    assert np.abs(out["force"] - ref["force"]).max() < 1e-3
    assert np.abs(out["virial"] - ref["virial"]).max() < 1e-2
    ani.close()


def test_build_list_argument_errors(model_cache, hip):
    import torch
    inp = hx.decompose(hx.random_box(300, 7, 16.0, seed=5))
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    with pytest.raises(hip.AniError, match="cutneigh"):
        _build(ani, inp, cutneigh=4.0)
    x = torch.zeros((inp.ntotal, 3), dtype=torch.float64, device="cuda:0")
    f = torch.zeros_like(x)
    ev = torch.zeros(10, dtype=torch.float64, device="cuda:0")
    with pytest.raises(hip.AniError, match="ago != 0"):  # no list installed yet
        ani.compute_device(inp.ntotal, inp.nlocal, None, x.data_ptr(), 0, None, None, None, 1, f.data_ptr(), ev.data_ptr())
    half = hip.ANI(model_cache("ani2x", 1, 11), 0, use_fullnbr=False)
    with pytest.raises(hip.AniError, match="full list"):
        _build(half, inp)
    ani.close()
    half.close()


def _md(hip, tmp_path, natoms, single, steps, dt=0.25, sort=True, overlap=None):
    import torch
    from lammps_ani_amd import md
    path = str(tmp_path / "gentle.anim")
    # output layer at 0.02: the random surface has no minimum at the start structure; this keeps it within a few kT
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    sysm = hx.water_box(natoms)
    if sort:
        sysm = hx.spatial_sort(sysm)
    inp = hx.decompose(sysm)
    ani = hip.ANI(path, 0, use_single=single)
    run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, torch.device("cuda:0"), dt=dt, box_lo=sysm.boxlo, overlap=overlap)
    run.create_velocities(300.0)
    e = [(run.potential_energy(), run.kinetic_energy())]
    for _ in range(steps):
        run.step()
        e.append((run.potential_energy(), run.kinetic_energy()))
    e = np.array(e)
    builds = run.nbuilds
    ani.close()
    return e, builds


@pytest.mark.parametrize("single", [True, False], ids=["fp32", "fp64"])
def test_nve_conserves_energy_through_rebuilds(single, tmp_path, hip):
    e, builds = _md(hip, tmp_path, 1536, single, 150)
    etot = e.sum(1)
    ke_change = abs(e[-1, 1] - e[0, 1])
    drift = np.abs(etot - etot[0]).max()
    print(f"builds {builds}, KE change {ke_change:.2f}, max |E - E0| {drift:.4f} kcal/mol")
    assert builds >= 3                 # the run crossed several re-neighbourings (wrap + ghost regeneration)
    assert ke_change > 50.0            # energy really flowed between potential and kinetic
    assert drift < 0.03                # ... and the sum stayed put (observed 0.007; dt = 0.25 fs integration error)


def test_overlapped_exchange_follows_the_plain_loop(tmp_path, hip):
    """The loop with its ghost exchanges on a second stream beside the rows without ghosts (the library's split step)
    is the same dynamics: NVE from the same start follows the plain loop -- energies step by step while the trajectories
    are still the same to rounding, conservation throughout, re-neighbouring at the same steps."""
    e0, b0 = _md(hip, tmp_path, 1536, True, 120)
    e1, b1 = _md(hip, tmp_path, 1536, True, 120, overlap=True)
    assert b0 == b1 and b0 >= 2
    assert np.abs(e1[:40] - e0[:40]).max() < 2e-3          # fp32 atomics order only; chaos has not amplified it yet
    assert np.abs(e1.sum(1) - e1.sum(1)[0]).max() < 0.03


def test_md_loop_raises_on_capacity_overflow(tmp_path, hip):
    """The device entry point cannot return ANI_ERR_CAPACITY (nothing synchronises): the energy turns into NaN and the
    loop's displacement check, which is a host round trip anyway, raises.  Provoked with atoms piled inside Rca."""
    import torch
    from lammps_ani_amd import md
    path = str(tmp_path / "gentle.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    rng = np.random.default_rng(0)
    n = 400   # 400 atoms in a 6 A blob of a 30 A box: > 96 neighbours inside 3.5 A
    x = 15.0 + rng.normal(0.0, 1.2, size=(n, 3))
    sysm = hx.System(x, np.full(n, 1, np.int32), np.zeros(3), np.full(3, 30.0))
    inp = hx.decompose(sysm)
    ani = hip.ANI(path, 0)
    with pytest.raises(RuntimeError, match="capacity|non-finite"):
        run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, torch.device("cuda:0"), dt=0.01, box_lo=sysm.boxlo, every=1)
        for _ in range(3):
            run.step()
    ani.close()


def test_ghost_fold_and_fused_integrator_follow_the_plain_loop(tmp_path, hip, monkeypatch):
    """One rank: (a) the library's pack / finish kernels doing the two ghost exchanges themselves (ani_set_ghost_fold) against
    the loop's own forward / reverse ghost kernels; (b) `run(N)` — final_integrate of a step and initial_integrate of the next
    as one kernel — against N calls of `step()`.  Same arithmetic; the forces differ by the order of fp32 atomics only.  60
    steps at 0.25 fs with hot start velocities: the loop re-neighbours on the way (the maps of a fold belong to a list epoch)."""
    import torch
    from lammps_ani_amd import md
    path = str(tmp_path / "gentle.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    sysm = hx.spatial_sort(hx.water_box(1536))
    inp = hx.decompose(sysm)
    table = np.random.default_rng(99).normal(0.0, 0.03, size=(sysm.natoms, 3))
    dev = torch.device("cuda:0")

    def trajectory(fold, fused_run, langevin=None):
        monkeypatch.setenv("ANI_MD_FOLD", "1" if fold else "0")
        ani = hip.ANI(path, 0)
        run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=0.25, box_lo=sysm.boxlo, langevin=langevin)
        assert run._fold == fold
        run.v = torch.as_tensor(table[run.tag.cpu().numpy()], dtype=torch.float64, device=dev)
        if fused_run:
            run.run(25)
            run.run(1)
            run.run(34)
        else:
            for _ in range(60):
                run.step()
        x = np.zeros((sysm.natoms, 3))
        x[run.tag.cpu().numpy()] = run.x[: run.nlocal].cpu().numpy()
        v = np.zeros((sysm.natoms, 3))
        v[run.tag.cpu().numpy()] = run.v.cpu().numpy()
        out = (x, v, run.potential_energy() + run.kinetic_energy(), run.nbuilds)
        ani.close()
        return out

    base = trajectory(False, False)
    assert base[3] >= 2                                  # it re-neighboured
    for fold, fused in ((True, False), (False, True), (True, True)):
        x, v, e, nb = trajectory(fold, fused)
        assert nb == base[3]
        assert np.abs(x - base[0]).max() < 1e-5 and np.abs(v - base[1]).max() < 1e-5, (fold, fused)
        assert abs(e - base[2]) < 5e-3
    # with the thermostat: the fused kernel draws the same random numbers (keyed by seed, step, tag) as final_integrate
    a = trajectory(False, False, langevin=(300.0, 100.0))
    b = trajectory(True, True, langevin=(300.0, 100.0))
    assert np.abs(a[0] - b[0]).max() < 1e-5 and np.abs(a[1] - b[1]).max() < 1e-5


def test_native_reneighbouring_equals_the_tensor_form(tmp_path, hip, monkeypatch):
    """One rank: position wrap, ghost shell (count / scan / fill), appended ghosts and the displacement check through the kernels
    of include/ani_md.h against the tensor operations of comm.DomainComm — the same ghost shell in the same order (combination-
    major, atoms ascending), hence the same device list and, to the fp32 atomics, the same trajectory.  A small box (three
    images per dimension are in reach of some atoms) and the benchmark-like one."""
    import torch
    from lammps_ani_amd import md
    path = str(tmp_path / "gentle.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 1, seed=1, out_scale=0.02))
    dev = torch.device("cuda:0")
    for natoms, steps in ((192, 30), (3000, 50)):
        sysm = hx.spatial_sort(hx.water_box(natoms))
        inp = hx.decompose(sysm)
        table = np.random.default_rng(7).normal(0.0, 0.04, size=(sysm.natoms, 3))
        res = {}
        for native in (False, True):
            monkeypatch.setenv("ANI_MD_NATIVE_REBUILD", "1" if native else "0")
            ani = hip.ANI(path, 0)
            run = md.VerletRun(ani, inp, sysm.boxhi - sysm.boxlo, dev, dt=0.25, box_lo=sysm.boxlo)
            assert run._native_rebuild == native
            run.v = torch.as_tensor(table[run.tag.cpu().numpy()], dtype=torch.float64, device=dev)
            shells = [(run.ntotal, run.npairs, run.dc.send_idx.cpu().numpy().copy(), run.dc.send_shift.cpu().numpy().copy())]
            for k in range(steps):
                nb = run.nbuilds
                run.step(force_rebuild=(k == 3))
                if run.nbuilds != nb:
                    shells.append((run.ntotal, run.npairs, run.dc.send_idx.cpu().numpy().copy(), run.dc.send_shift.cpu().numpy().copy()))
            x = np.zeros((sysm.natoms, 3))
            x[run.tag.cpu().numpy()] = run.x[: run.nlocal].cpu().numpy()
            res[native] = (shells, x, run.potential_energy() + run.kinetic_energy())
            ani.close()
        a, b = res[False], res[True]
        assert len(a[0]) == len(b[0]) >= 2
        # the first shells (same start positions) are identical entry by entry; later ones only as long as no atom sits within the
        # trajectories' fp32 noise of a shell boundary, so their sizes are compared
        assert a[0][0][0] == b[0][0][0] and a[0][0][1] == b[0][0][1]
        assert np.array_equal(a[0][0][2], b[0][0][2]) and np.array_equal(a[0][0][3], b[0][0][3])
        for sa, sb in zip(a[0][1:], b[0][1:]):
            assert abs(sa[0] - sb[0]) <= 2 and abs(sa[1] - sb[1]) <= 40
        assert np.abs(a[1] - b[1]).max() < 1e-5 and abs(a[2] - b[2]) < 5e-3


def _species_of(inp):
    return np.asarray(inp.species)


@pytest.mark.parametrize("half_cells", [0, 1], ids=["cells_cutoff", "cells_half_cutoff"])
def test_one_kernel_build_equals_search_then_sort(half_cells, model_cache, hip):
    """The build in one kernel (fp32 screening with the fp64 decision near the cutoff, rows grouped by species on the way out)
    against the separate search and grouping kernels: the same entries in the same ORDER, centre by centre; grouped by species;
    and the same sets as the harness's host build.  Second build of a handle: the first has no row capacity to go by."""
    for sysm, grid, rank in ((hx.spatial_sort(hx.water_box(3000)), (1, 1, 1), 0), (hx.random_box(2500, 7, 31.0, seed=4), (2, 2, 2), 5),
                             (hx.random_box(300, 7, 15.0, seed=5), (1, 1, 1), 0)):
        inp = hx.decompose(sysm, grid=grid, rank=rank)
        lists = {}
        for rows in (0, 1):
            ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
            ani.set_option("nbr_sorted_rows", rows)
            ani.set_option("nbr_half_cells", half_cells if rows else 0)
            for pad in (0.25, -3.0):
                _build(ani, inp, pad=pad)
                n, _, _ = _build(ani, inp, pad=pad)       # the build that has a capacity
                assert n == inp.npairs
                lists[(rows, pad)] = ani.debug_list(inp.nlocal)
            ani.close()
        sp = _species_of(inp)
        for pad in (0.25, -3.0):
            (nn0, jl0), (nn1, jl1) = lists[(0, pad)], lists[(1, pad)]
            assert np.array_equal(nn0, inp.numneigh) and np.array_equal(nn1, inp.numneigh)
            if not half_cells:
                assert np.array_equal(jl0, jl1)            # same serial order inside every species group
            off = np.concatenate([[0], np.cumsum(nn1)])
            for i in range(inp.nlocal):
                seg = jl1[off[i]:off[i + 1]]
                assert (np.diff(sp[seg]) >= 0).all()       # grouped by species, ascending
            for a, b in zip(_segments(nn1, jl1), _segments(inp.numneigh, inp.jlist)):
                assert np.array_equal(a, b)


def test_one_kernel_build_decides_pairs_at_the_cutoff_in_fp64(model_cache, hip):
    """Pairs placed a few parts in 1e9 inside and outside the neighbour cutoff -- far inside the band in which the fp32 screening
    cannot tell -- far from the origin of the grid (where fp32 is at its worst): the list must be the fp64 brute-force list."""
    rng = np.random.default_rng(5)
    cut, n = 7.1, 400
    base = rng.uniform(0.0, 60.0, (n, 3)) + 400.0           # 400 A from the grid origin (lo is set far away below)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    eps = rng.choice([-4e-9, -1e-9, 1e-9, 4e-9], size=n)
    x = np.concatenate([base, base + u * (cut * (1.0 + eps))[:, None]])
    ntotal = 2 * n
    species = rng.integers(0, 7, ntotal).astype(np.int32)
    d = x[:, None, :] - x[None, :, :]
    r2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    want = (r2 <= cut * cut) & ~np.eye(ntotal, dtype=bool)
    near = np.abs(np.sqrt(r2) - cut) < 1e-7
    assert near.sum() >= 2 * n                               # the constructed pairs, both directions
    import torch
    dev = torch.device("cuda:0")
    xd = torch.as_tensor(x, device=dev).contiguous()
    sd = torch.as_tensor(species, device=dev)
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    lo, hi = np.zeros(3), x.max(0) + 1.0
    for rows in (0, 1):
        ani.set_option("nbr_sorted_rows", rows)
        for _ in range(2):
            ani.build_list_device(ntotal, ntotal, sd.data_ptr(), xd.data_ptr(), cut, lo, hi)
        torch.cuda.synchronize()
        nn, jl = ani.debug_list(ntotal)
        off = np.concatenate([[0], np.cumsum(nn)])
        # pairs within one ulp of the fp64 sum may fall either way (fused multiply-adds on the device): none were constructed
        for i in range(ntotal):
            assert np.array_equal(np.sort(jl[off[i]:off[i + 1]]), np.nonzero(want[i])[0]), (rows, i)
    ani.close()


def test_one_kernel_build_falls_back_when_a_row_overflows(model_cache, hip):
    """Rows sized from a sparse system, then a dense one through the same handle: the overflow word sends the build to the
    count-and-fill path, and the list is still the host list; the build after that has rows again."""
    sparse = hx.decompose(hx.random_box(600, 7, 40.0, seed=2))
    dense = hx.decompose(hx.random_box(1200, 7, 20.0, seed=3))
    assert dense.numneigh.max() > 2 * sparse.numneigh.max() + 16
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    _build(ani, sparse)
    _build(ani, sparse)
    for _ in range(2):
        n, _, _ = _build(ani, dense)
        assert n == dense.npairs
        nn, jl = ani.debug_list(dense.nlocal)
        assert np.array_equal(nn, dense.numneigh)
        for a, b in zip(_segments(nn, jl), _segments(dense.numneigh, dense.jlist)):
            assert np.array_equal(a, b)
    ani.close()


def test_staged_ghost_fold_is_checked_with_the_list(model_cache, hip):
    import torch
    inp = hx.decompose(hx.spatial_sort(hx.water_box(900)))
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    dev = torch.device("cuda:0")
    ng = inp.ntotal - inp.nlocal
    owner = torch.as_tensor(inp.owner_lidx.astype(np.int64), device=dev)
    shift = torch.zeros((ng, 3), dtype=torch.float64, device=dev)
    bad = owner.clone()
    bad[ng // 2] = inp.nlocal + 5
    for _ in range(2):                                       # both build paths
        ani.stage_ghost_fold(bad.data_ptr(), shift.data_ptr(), ng)
        with pytest.raises(hip.AniError, match="owner index"):
            _build(ani, inp)
    ani.stage_ghost_fold(owner.data_ptr(), shift.data_ptr(), ng)
    _build(ani, inp)
    ani.stage_ghost_fold(owner.data_ptr(), shift.data_ptr(), ng + 1)   # not this list's ghost count: dropped, no error
    _build(ani, inp)
    ani.close()
