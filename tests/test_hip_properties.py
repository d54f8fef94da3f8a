"""GPU tests beyond the fixtures: size-independent properties at the benchmark's full size, rank-count invariance of
the HIP path, the fp32-vs-fp64 force sweep on mixed-species boxes (BASELINE.json configs[4]: reactive C/H/N/O box,
"fp32 vs fp64 force tolerance sweep"), and the error paths of the C ABI."""
import os

import numpy as np
import pytest

from lammps_ani_amd import harness as hx

pytestmark = pytest.mark.gpu

F_TOL = 2.3e-3  # kcal/mol/A = 1e-4 eV/A (north-star bar)


@pytest.fixture(scope="module")
def hip():
    from lammps_ani_amd import ani_hip
    return ani_hip


def _folded(inp, force):
    f = force[: inp.nlocal].copy()
    np.add.at(f, inp.owner_lidx, force[inp.nlocal:])
    return f


def test_full_size_water_box_properties(model_cache, hip):
    """The 100 002-atom benchmark box (too big for the oracle inside a unit test): properties that hold at any size.
      * Newton's third law: forces folded onto owners sum to zero (fp32 noise of 1e5 atoms);
      * translation invariance: shifting every atom by the same vector leaves energy and forces unchanged;
      * the virial equals sum_i r_i (x) f_i over local+ghost atoms (the reference's identity for a potential that
        depends on differences only), which ties the separately accumulated virial to the forces."""
    p = model_cache("ani2x", 1, 2024)
    sysm = hx.spatial_sort(hx.water_box(100002))
    inp = hx.decompose(sysm)
    ani = hip.ANI(p, 0)
    a = ani.compute(inp, ago=0)
    fa = _folded(inp, a["force"])
    assert np.isfinite(a["energy"])
    assert np.abs(fa.sum(0)).max() < 0.5                       # |sum F| over 1e5 atoms with rms force ~16
    assert 5.0 < np.sqrt((fa ** 2).mean()) < 50.0
    shifted = hx.RankInput(**{**inp.__dict__, "x": inp.x + np.array([0.37, -1.21, 2.5])})
    b = ani.compute(shifted, ago=1)
    assert abs(b["energy"] - a["energy"]) < 0.05               # 4.8e7 kcal/mol total, fp32 per-atom terms
    assert np.abs(b["force"] - a["force"]).max() < F_TOL
    w = (inp.x[:, :, None] * a["force"][:, None, :]).sum(0)    # sum r (x) f, all atoms incl. ghosts
    w = 0.5 * (w + w.T)
    assert np.abs(w - a["virial"]).max() < 2e-3 * np.abs(a["virial"]).max() + 1.0
    ani.close()


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)], ids=["2ranks", "8ranks"])
def test_rank_count_invariance_on_device(grid, model_cache, hip):
    """LAMMPS' brick decomposition: the per-rank energies sum to the single-rank energy and the folded forces agree."""
    p = model_cache("ani2x", 2, 2024)
    sysm = hx.random_box(1800, 7, 30.0, seed=12)
    one = hx.decompose(sysm)
    ani = hip.ANI(p, 0)
    r1 = ani.compute(one, ago=0)
    f1 = np.zeros((sysm.natoms, 3))
    f1[one.tag[: one.nlocal]] = _folded(one, r1["force"])
    e, F = 0.0, np.zeros((sysm.natoms, 3))
    vir = np.zeros((3, 3))
    for rank in range(int(np.prod(grid))):
        inp = hx.decompose(sysm, grid, rank)
        r = ani.compute(inp, ago=0)
        e += r["energy"]
        vir += r["virial"]
        np.add.at(F, inp.tag, r["force"])      # ghost rows land on their owners' global index
    assert abs(e - r1["energy"]) < 2e-2
    assert np.abs(F - f1).max() < F_TOL
    assert np.abs(vir - r1["virial"]).max() < 0.2
    ani.close()


@pytest.mark.parametrize("natoms,L", [(600, 22.0), (1200, 24.0), (2400, 27.0), (4000, 30.0)])
def test_fp32_vs_fp64_force_sweep_mixed_species(natoms, L, model_cache, hip):
    """C/H/N/O boxes of growing density (0.056 .. 0.148 atoms/A^3, up to ~75 radial neighbours per atom): the fp32
    MFMA path against the fp64 kernels of the same library."""
    p = model_cache("ani1x", 4, 77)
    inp = hx.decompose(hx.random_box(natoms, 4, L, seed=natoms))
    a32 = hip.ANI(p, 0, use_single=True)
    a64 = hip.ANI(p, 0, use_single=False)
    r32, r64 = a32.compute(inp, ago=0), a64.compute(inp, ago=0)
    df = np.abs(r32["force"] - r64["force"])
    print(f"natoms {natoms}: max|dF| {df.max():.2e} rms {np.sqrt((df ** 2).mean()):.2e} max|F| {np.abs(r64['force']).max():.1f}")
    assert df.max() < F_TOL
    assert abs(r32["energy"] - r64["energy"]) < 2e-3 * max(1.0, natoms / 100.0)
    assert np.abs(r32["eatom"] - r64["eatom"]).max() < 2e-3
    a32.close()
    a64.close()


def test_neighbour_capacity_is_reported_not_ignored(model_cache, hip):
    """More than 96 neighbours inside the 3.5 A angular cutoff (a density no real system has): the step must fail with
    ANI_ERR_CAPACITY, never return numbers computed from a truncated neighbourhood."""
    p = model_cache("ani2x", 1, 2024)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1.6, 1.6, size=(130, 3))              # 130 atoms in a 3.2 A cube
    s = hx.System(x, np.full(130, 1, np.int32), np.full(3, -30.0), np.full(3, 30.0), (False,) * 3)
    inp = hx.decompose(s)
    ani = hip.ANI(p, 0)
    with pytest.raises(hip.AniError, match="capacity"):
        ani.compute(inp, ago=0)
    # the handle stays usable
    ok = hx.decompose(hx.water_box(300, seed=2))
    assert np.isfinite(ani.compute(ok, ago=0)["energy"])
    ani.close()


def test_argument_errors(model_cache, hip):
    p = model_cache("tiny", 2, 5)   # 3 species
    ani = hip.ANI(p, 0)
    inp = hx.decompose(hx.random_box(60, 3, 14.0, seed=1))
    with pytest.raises(hip.AniError, match="ago != 0"):
        ani.compute(inp, ago=1)                              # no list cached yet
    bad = hx.RankInput(**{**inp.__dict__, "types": np.where(np.arange(inp.ntotal) == 5, 9, inp.types).astype(np.int32)})
    with pytest.raises(hip.AniError, match="species"):
        ani.compute(bad, ago=0)
    good = ani.compute(inp, ago=0)
    assert np.isfinite(good["energy"])
    with pytest.raises(hip.AniError):
        hip.ANI(p, 0, use_num_models=5)                      # the file holds 2 members
    ani.close()


def test_split_mlp_arithmetics_are_as_accurate_as_fp32_input_mfma(model_cache, hip):
    """The MLP evaluates its fp32 products on the 16-bit MFMA pipe (option mlp_arith): 2 (opt-in) = three fp16 products of
    two-term splits of the power-of-two scaled operands (operands good to 2^-22), 1 (default) = six bf16 products of the exact
    three-term splits, 0 = the fp32-input MFMA instruction.  Against the fp64 oracle each split path must be as good as
    the fp32-input one -- forces AND per-atom energies, max and rms -- and all must agree with each other far inside the
    force tolerance; water with one member (chained launch and per-layer launches) and the full 8-member ensemble on a
    mixed-species box."""
    from oracle import Oracle
    for kind, nm, inp, chain in (("ani2x", 1, hx.decompose(hx.water_box(1500, seed=5)), 1),
                                 ("ani2x", 1, hx.decompose(hx.water_box(1500, seed=5)), 0),
                                 ("ani2x", 8, hx.decompose(hx.random_box(700, 7, 22.0, seed=9)), 1)):
        p = model_cache(kind, nm, 2024)
        ref = Oracle(p).compute(inp)
        ani = hip.ANI(p, 0)
        ani.set_option("mlp_chain", chain)
        out, err, rms, eat = {}, {}, {}, {}
        for arith in (2, 1, 0):
            ani.set_option("mlp_arith", arith)
            out[arith] = ani.compute(inp, ago=0)
            d = out[arith]["force"] - ref["force"]
            err[arith], rms[arith] = np.abs(d).max(), np.sqrt((d ** 2).mean())
            eat[arith] = np.abs(out[arith]["eatom"] - ref["eatom"][: inp.nlocal]).max()
        print(f"{kind} x{nm} chain={chain}: max|dF| f16x2 {err[2]:.2e} bf16x3 {err[1]:.2e} fp32 {err[0]:.2e}; rms {rms[2]:.2e} {rms[1]:.2e} "
              f"{rms[0]:.2e}; max|dE_atom| {eat[2]:.2e} {eat[1]:.2e} {eat[0]:.2e}")
        for arith in (2, 1):
            assert err[arith] < F_TOL and err[0] < F_TOL
            assert err[arith] < 1.5 * err[0] + 1e-5 and rms[arith] < 1.25 * rms[0] + 1e-6
            assert eat[arith] < 1.5 * eat[0] + 2e-5
            assert np.abs(out[arith]["force"] - out[0]["force"]).max() < 0.25 * F_TOL
            assert abs(out[arith]["energy"] - out[0]["energy"]) < 1e-3 * max(1.0, inp.nlocal / 100.0)
        ani.close()


def test_fp16_split_overflow_is_loud(model_cache, hip, tmp_path):
    """The two-term fp16 path scales activations by 2^4: a hidden activation beyond 4094 is outside fp16's range and must
    surface as a non-finite energy (which the entry points report), never as a finite wrong one.  Built by blowing up a
    first-layer bias of a synthetic model; the exact bf16 path evaluates the same model finitely."""
    from lammps_ani_amd import model_file as mf
    m = mf.synthetic_model("ani2x", 1, seed=2024)
    for s in range(len(m.species)):
        m.weights[0][s][0][1][:] = 1.0e4        # bias of layer 0: every activation of layer 1 ~ 1e4
    p = str(tmp_path / "big.anim")
    mf.write_model(p, m)
    inp = hx.decompose(hx.water_box(600, seed=5))
    ani = hip.ANI(p, 0)
    ani.set_option("mlp_arith", 1)
    assert np.isfinite(ani.compute(inp, ago=0)["energy"])
    ani.set_option("mlp_arith", 2)
    try:
        e = ani.compute(inp, ago=0)["energy"]
    except hip.AniError:
        e = float("nan")
    assert not np.isfinite(e)
    ani.close()


def test_screened_radial_capacity_overflow_is_loud_and_option_restores_it(model_cache, hip):
    """With the radial screen on, LDS is reserved for 3/4 of the longest neighbour list.  A cluster in vacuum whose
    whole 7.1 A list sits inside 5.1 A (the one geometry where that estimate is wrong) must raise the capacity error,
    and option full_radial_capacity must then give the oracle's answer."""
    from oracle import Oracle
    p = model_cache("ani2x", 1, 2024)
    rng = np.random.default_rng(11)
    pts = []
    while len(pts) < 240:                                   # 240 atoms in a 5 A ball: every atom sees >128 others < 5.1 A
        q = rng.uniform(-2.5, 2.5, 3)
        if q @ q < 2.5 ** 2 and all(np.sum((q - r) ** 2) > 0.55 ** 2 for r in pts):
            pts.append(q)
    s = hx.System(np.array(pts), np.full(240, 1, np.int32), np.full(3, -30.0), np.full(3, 30.0), (False,) * 3)
    inp = hx.decompose(s)
    assert inp.numneigh.max() > 171                         # 3/4 of it rounds to 192 slots at most
    ani = hip.ANI(p, 0)
    with pytest.raises(hip.AniError, match="capacity"):
        ani.compute(inp, ago=0)
    ani.close()
    # the angular capacity (96 within 3.5 A) also overflows for this blob, so check the option on a sparser ball
    shell = np.array(pts) * 3.0                              # 7.5 A radius: <= ~80 atoms within 5.1 A, lists of ~200
    s2 = hx.System(shell, np.full(240, 1, np.int32), np.full(3, -40.0), np.full(3, 40.0), (False,) * 3)
    inp2 = hx.decompose(s2)
    ani = hip.ANI(p, 0)
    ani.set_option("full_radial_capacity", 1)
    got = ani.compute(inp2, ago=0)
    ref = Oracle(p).compute(inp2)
    assert np.abs(got["force"] - ref["force"]).max() < F_TOL
    ani.set_option("full_radial_capacity", 0)
    got2 = ani.compute(inp2, ago=0)
    assert np.abs(got2["force"] - got["force"]).max() < 1e-4
    ani.close()


def test_repulsion_term_rank_invariance_and_oracle(model_cache, hip):
    """Optional pairwise repulsion (model-file block REPULXTB): an owned-ghost pair counts half on each rank, so the
    per-rank energies still sum to the single-rank energy; forces and virial against the oracle in both precisions."""
    from oracle import Oracle
    p = model_cache("ani1x", 2, 7, True)
    p0 = model_cache("ani1x", 2, 7)
    sysm = hx.random_box(900, 4, 24.0, seed=3, min_dist=1.0)
    one = hx.decompose(sysm)
    ref = Oracle(p).compute(one)
    for single in (True, False):
        ani = hip.ANI(p, 0, use_single=single)
        r1 = ani.compute(one, ago=0)
        tol = F_TOL if single else 1e-7
        assert np.abs(r1["force"] - ref["force"]).max() < tol
        assert abs(r1["energy"] - ref["energy"]) < (2e-2 if single else 1e-6)
        assert np.abs(r1["virial"] - ref["virial"]).max() < (0.2 if single else 1e-6)
        e = sum(ani.compute(hx.decompose(sysm, (2, 2, 1), rank), ago=0)["energy"] for rank in range(4))
        assert abs(e - r1["energy"]) < (2e-2 if single else 1e-6)
        ani.close()
    # the block really adds something: same weights without it differ
    base = hip.ANI(p0, 0)
    assert abs(base.compute(one, ago=0)["energy"] - ref["energy"]) > 1.0
    base.close()


def test_chained_mlp_launch_equals_per_layer_launches(model_cache, hip):
    """Small single-member systems run the six MLP products as one chained launch (a workgroup carries its row tile through
    all layers); option mlp_chain = 0 forces the per-layer launches.  Same arithmetic up to the summation order over k
    (the per-layer kernel rotates its K start per workgroup) and the order of the force atomics: agreement far inside the
    tolerance.  Sizes below and above the switch from 32- to 64-row tiles, water (pruned AEV) and a 7-species box."""
    p = model_cache("ani2x", 1, 2024)
    for inp in (hx.decompose(hx.water_box(3000, seed=3)), hx.decompose(hx.water_box(24000, seed=4)),
                hx.decompose(hx.random_box(1200, 7, 26.0, seed=6))):
        ani = hip.ANI(p, 0)
        a = ani.compute(inp, ago=0)
        ani.set_option("mlp_chain", 0)
        b = ani.compute(inp, ago=0)
        assert np.abs(a["force"] - b["force"]).max() < 0.1 * F_TOL
        assert abs(a["energy"] - b["energy"]) < 1e-3 * max(1.0, inp.nlocal / 1000.0)
        assert np.abs(a["eatom"] - b["eatom"]).max() < 1e-4
        ani.close()


def test_symmetric_radial_collection_equals_the_scatter(model_cache, hip):
    """Option aev_symmetric_radial (default 1): a centre takes both radial terms of a pair with a local neighbour on itself (the
    neighbour's read from the neighbour's dE/dAEV row) and scatters no radial gradient to it; 0 = every radial gradient is
    scattered to the neighbour.  The same forces, energy and virial: water, a 7-species box, an ANI-1x box with the pairwise
    repulsion (whose pair term is split between the two entries of a pair), each with its periodic ghosts (which have no row and
    keep the scatter)."""
    cases = [("ani2x", hx.decompose(hx.water_box(3000, seed=3)), False), ("ani2x", hx.decompose(hx.random_box(1200, 7, 26.0, seed=6)), False),
             ("ani1x", hx.decompose(hx.random_box(900, 4, 24.0, seed=8), cutoff=5.2), True)]
    for kind, inp, rep in cases:
        p = model_cache(kind, 2, 79, repulsion=rep)
        ani = hip.ANI(p, 0)
        out = []
        for sym in (1, 0):
            ani.set_option("aev_symmetric_radial", sym)
            out.append(ani.compute(inp, ago=0, eflag_atom=True, vflag=True))
        fmax = max(1.0, float(np.abs(out[1]["force"]).max()))
        assert np.abs(out[0]["force"] - out[1]["force"]).max() < 0.02 * F_TOL * fmax
        assert abs(out[0]["energy"] - out[1]["energy"]) < 1e-5
        assert np.abs(out[0]["virial"] - out[1]["virial"]).max() < 2e-3 * max(1.0, inp.nlocal / 100.0)
        ani.close()


def test_rows_by_ticket_equal_rows_by_stride(model_cache, hip):
    """Option aev_tickets_min: large launches hand the AEV rows to the waves by ticket (contiguous blocks per group of workgroups
    in the forward launch, the groups' rows interleaved in the backward launch), small ones at a fixed stride.  Forced on for
    systems far below the default threshold -- fewer rows than ticket groups, row counts that no group count divides, both
    forward kernels -- every row must be computed exactly once: AEV rows equal bit by bit, forces to the order of the atomics."""
    cases = [("ani2x", hx.decompose(hx.water_box(30, seed=2)), 1), ("ani2x", hx.decompose(hx.water_box(1001 * 3, seed=3)), 1),
             ("ani2x", hx.decompose(hx.random_box(1237, 7, 26.0, seed=6)), 1), ("ani2x", hx.decompose(hx.water_box(6000, seed=5)), 0)]
    for kind, inp, fused in cases:
        p = model_cache(kind, 2, 78)
        ani = hip.ANI(p, 0)
        ani.set_option("aev_fused", fused)
        out, rows = [], []
        for tmin in (0, 1 << 30):
            ani.set_option("aev_tickets_min", tmin)
            for _ in range(2):   # twice: the counters must be back at zero after a launch
                o = ani.compute(inp, ago=0)
            out.append(o)
            v = ani.debug_view()
            rows.append(ani.debug_read(v.d_aev, (v.nrows, v.aev_stride), np.float32))
        assert np.array_equal(rows[0], rows[1])
        assert abs(out[0]["energy"] - out[1]["energy"]) < 1e-5
        assert np.abs(out[0]["force"] - out[1]["force"]).max() < 0.02 * F_TOL
        ani.close()


def test_compaction_inside_the_forward_launch_equals_the_two_kernels(model_cache, hip):
    """Option aev_fused (default 1): the wave that featurises a centre screens its candidate list itself; 0 = the compaction
    kernel in front of the forward kernel.  Both fill the same LDS lists in the same order, so the AEV rows must be equal BIT BY
    BIT and the compact lists the backward kernel starts from too (forces differ by the order of the fp32 force atomics only).
    Water (pruned AEV), a 7-species box, an ANI-1x shaped model (the other kernel instantiation) and the unscreened radial
    list (use_cuaev = False: every candidate stays, three chunks per centre)."""
    cases = [("ani2x", hx.decompose(hx.water_box(6000, seed=3)), True), ("ani2x", hx.decompose(hx.random_box(1200, 7, 26.0, seed=6)), True),
             ("ani1x", hx.decompose(hx.random_box(900, 4, 24.0, seed=8), cutoff=5.2), True), ("ani2x", hx.decompose(hx.water_box(3000, seed=5)), False)]
    for kind, inp, cuaev in cases:
        p = model_cache(kind, 2, 77)
        ani = hip.ANI(p, 0, use_cuaev=cuaev)
        out, rows = [], []
        for fused in (1, 0):
            ani.set_option("aev_fused", fused)
            out.append(ani.compute(inp, ago=0))
            v = ani.debug_view()
            rows.append(ani.debug_read(v.d_aev, (v.nrows, v.aev_stride), np.float32))
        assert np.array_equal(rows[0], rows[1])
        assert abs(out[0]["energy"] - out[1]["energy"]) < 1e-5   # the same rows through the same MLP: the order of the energy sum only
        assert np.abs(out[0]["force"] - out[1]["force"]).max() < 0.02 * F_TOL
        ani.close()


@pytest.mark.skipif(not os.environ.get("ANI_TEST_PIPELINE"), reason="the one-launch pipeline is an opt-in experiment since round 4 "
                    "(mlp_pipeline default 0): one suite run in three saw a stale tile at 60 000 atoms, step 4; ANI_TEST_PIPELINE=1 runs it")
@pytest.mark.parametrize("natoms", [3000, 24000, 60000])
def test_mlp_pipeline_equals_per_layer_launches_on_changing_inputs(natoms, model_cache, hip):
    """Large single-member systems run all MLP layers as ONE launch of persistent workgroups: tile t of layer l waits for a
    completion flag of tile t of layer l - 1, whose results another workgroup -- possibly on another XCD, whose L2 is not
    coherent with this one's inside a launch -- stored write-through.  A stale read would return the PREVIOUS step's
    activations (the buffers are reused every step), which a benchmark on static positions cannot see: here the atoms
    move between steps, and every step must match a handle that launches layer by layer.  Option mlp_pipeline = 2 forces
    the path at sizes where it is not the default (fewer items than resident workgroups, about as many, several times)."""
    p = model_cache("ani2x", 1, 2024)
    sysm = hx.spatial_sort(hx.water_box(natoms, seed=11))
    inp = hx.decompose(sysm)
    piped, layered = hip.ANI(p, 0), hip.ANI(p, 0)
    for h, v in ((piped, 2), (layered, 0)):
        h.set_option("mlp_chain", 0)
        h.set_option("mlp_arith", 2)      # the one-launch pipeline exists for the two-term arithmetic (an opt-in since round 3)
        h.set_option("mlp_pipeline", v)
    rng = np.random.default_rng(3)
    direction = rng.normal(size=(inp.nlocal, 3))
    prev = None
    for step in range(5):
        disp = np.zeros_like(inp.x)
        disp[: inp.nlocal] = 0.03 * step * direction
        disp[inp.nlocal:] = disp[inp.owner_lidx]
        moved = hx.RankInput(**{**inp.__dict__, "x": inp.x + disp})
        a, b = piped.compute(moved, ago=step), layered.compute(moved, ago=step)
        assert np.isfinite(a["energy"])
        assert np.abs(a["force"] - b["force"]).max() < 2e-4, step
        assert np.abs(a["eatom"] - b["eatom"]).max() < 1e-4
        if prev is not None:   # the steps really differ: a stale activation would show
            assert np.abs(a["force"] - prev).max() > 0.05
        prev = a["force"]
    piped.close(); layered.close()


def test_energy_is_extensive_under_periodic_replication(model_cache, hip):
    """Two periodic copies of a box side by side: E = 2 E(base) and every atom's force repeats -- a property that needs
    no oracle (tools/big_probe.py runs it with 8 copies = 10^6 atoms).  The copy sits a box length away from the origin:
    with absolute fp32 coordinates its rounding would differ from the base box's; the epoch origin keeps them alike."""
    base = hx.spatial_sort(hx.water_box(6000, seed=21))
    L = base.boxhi - base.boxlo
    shift = np.array([L[0], 0.0, 0.0])
    hi2 = base.boxhi + shift
    tiled = hx.System(np.concatenate([base.x, base.x + shift]), np.tile(base.types, 2), base.boxlo, hi2)
    ani = hip.ANI(model_cache("ani2x", 2, 11), 0)
    def total_force(system):
        inp = hx.decompose(system)
        out = ani.compute(inp, ago=0)
        f = out["force"][: inp.nlocal].copy()
        np.add.at(f, inp.owner_lidx, out["force"][inp.nlocal:])   # ghost images' shares go home (the reverse communication)
        return out["energy"], f

    ea, fa = total_force(base)
    eb, fb = total_force(tiled)
    n = len(base.x)
    assert abs(eb - 2.0 * ea) < 1e-9 * abs(ea) + 2e-2
    assert np.abs(fb[:n] - fa).max() < 0.5 * F_TOL
    assert np.abs(fb[n:] - fa).max() < 0.5 * F_TOL
    ani.close()


def test_profiling_option_and_trace_markers(model_cache, hip):
    """LAMMPS_ANI_PROFILING / NVTX of the reference (src/pair_ani.cpp:49-50,198-200; src/ani_csrc/ani.cpp:128,215): option
    "profiling" makes the device entry point finish its work before returning (results unchanged), the roctx wrappers
    are callable with and without a profiler attached."""
    import torch
    lib = hip.lib()
    lib.ani_trace_mark.argtypes = [hip.C.c_char_p]
    lib.ani_trace_push.argtypes = [hip.C.c_char_p]
    lib.ani_trace_mark(b"test marker")
    lib.ani_trace_push(b"test range")
    lib.ani_trace_pop()
    inp = hx.decompose(hx.water_box(1536, seed=2))
    ani = hip.ANI(model_cache("ani2x", 1, 2024), 0)
    dev = torch.device("cuda:0")
    x = torch.as_tensor(inp.x, dtype=torch.float64, device=dev).contiguous()
    sp = torch.as_tensor(inp.species.astype(np.int32), device=dev)
    il, nn, jl = (torch.as_tensor(a, device=dev) for a in (inp.ilist, inp.numneigh, inp.jlist))
    out = []
    for prof in (0, 1):
        ani.set_option("profiling", prof)
        f = torch.zeros((inp.ntotal, 3), dtype=torch.float64, device=dev)
        ev = torch.zeros(10, dtype=torch.float64, device=dev)
        ani.compute_device(inp.ntotal, inp.nlocal, sp.data_ptr(), x.data_ptr(), inp.npairs, il.data_ptr(), jl.data_ptr(),
                           nn.data_ptr(), 0, f.data_ptr(), ev.data_ptr(), vflag=True)
        if prof:   # no synchronisation needed on our side: the call has already waited for its stream
            assert torch.cuda.current_stream().query()
        torch.cuda.synchronize()
        out.append((f.cpu().numpy(), ev.cpu().numpy()))
    assert np.abs(out[0][0] - out[1][0]).max() < F_TOL and abs(out[0][1][0] - out[1][1][0]) < 1e-3
    with pytest.raises(hip.AniError, match="unknown option"):
        ani.set_option("no_such_option", 1)
    ani.close()


def test_radial_capacity_overflow_is_retried_with_full_capacity(model_cache, hip, capfd):
    """A geometry whose list sits entirely inside Rcr (atoms on a 2.5 A sphere: every pair closer than 5.1 A, about half
    of them inside Rca) overflows the screened radial capacity (3/4 of the list, at least 128) but not the angular one:
    the host entry point repeats the step with the full capacity instead of aborting the run, and says so once."""
    from oracle import Oracle
    p = model_cache("ani2x", 1, 2024)
    n = 140
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    th = np.pi * (1 + 5 ** 0.5) * k
    pts = 2.5 * np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    s = hx.System(pts, np.full(n, 1, np.int32), np.full(3, -30.0), np.full(3, 30.0), (False,) * 3)
    inp = hx.decompose(s)
    assert inp.numneigh.min() == n - 1
    ani = hip.ANI(p, 0)
    got = ani.compute(inp, ago=0)
    assert "full_radial_capacity = 1" in capfd.readouterr().err
    ref = Oracle(p).compute(inp)
    assert np.abs(got["force"] - ref["force"]).max() < 5e-2 and abs(got["energy"] - ref["energy"]) < 5e-2   # forces of 1e3, E 2e4 kcal/mol here
    got2 = ani.compute(inp, ago=1)           # the setting stays: no second retry, same answer
    assert "full_radial_capacity" not in capfd.readouterr().err
    assert np.abs(got2["force"] - got["force"]).max() < 1e-3
    ani.close()


def test_local_rank_maps_onto_the_visible_devices(model_cache, hip):
    """src/pair_ani.cpp:255-283: the node-local MPI rank is taken modulo the number of visible devices.  On a one-GPU box
    every rank lands on device 0 and computes the same thing; -1 (the reference's `device cpu`) is refused."""
    import torch
    p = model_cache("ani2x", 1, 2024)
    inp = hx.decompose(hx.water_box(300, seed=5))
    ndev = torch.cuda.device_count()
    ref = None
    for local_rank in (0, 1, 5, 8 * ndev + 3):
        ani = hip.ANI(p, local_rank)
        out = ani.compute(inp, ago=0)
        if local_rank % ndev == 0:
            if ref is None:
                ref = out
            # same device, same arithmetic; only the order of the fp32 force atomics differs between runs
            assert np.abs(out["force"] - ref["force"]).max() < 1e-4 and abs(out["energy"] - ref["energy"]) < 1e-4
        else:
            assert np.abs(out["force"] - ref["force"]).max() < F_TOL
        ani.close()
    with pytest.raises(hip.AniError, match="cpu"):
        hip.ANI(p, -1)


@pytest.mark.gpu
def test_out_force_accumulate_adds_in_chunks_and_nothing_on_error(model_cache, hip, monkeypatch):
    """Option out_force_accumulate: the host entry point ADDS the forces into the caller's array (three chunks here; one is the
    default) and leaves it alone when the call fails."""
    monkeypatch.setenv("ANI_FORCE_CHUNKS", "3")
    sysm = hx.spatial_sort(hx.water_box(21000))
    inp = hx.decompose(sysm)
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    ref = ani.compute(inp, ago=0)
    assert inp.ntotal > 3 * 8192
    rng = np.random.default_rng(0)
    pre = rng.normal(size=(inp.ntotal, 3))
    ani.set_option("out_force_accumulate", 1)
    f = pre.copy()
    out = ani.compute(inp, ago=1, force_into=f)
    assert out["force"] is f
    assert np.abs(f - (pre + ref["force"])).max() < 2e-5     # the force scatter's atomics are not ordered: last bits of fp32
    assert abs(out["energy"] - ref["energy"]) < 1e-6
    # a call that fails (capacity overflow with the retry exhausted is hard to stage: use the argument check) adds nothing
    g = pre.copy()
    with pytest.raises(hip.AniError):
        bad = hx.RankInput(**{**inp.__dict__, "nlocal": inp.nlocal - 1, "nghost": inp.nghost + 1})   # not the list's split
        ani.compute(bad, ago=1, force_into=g)
    assert np.array_equal(g, pre)
    ani.set_option("out_force_accumulate", 0)
    h = pre.copy()
    ani.compute(inp, ago=1, force_into=h)
    assert np.abs(h - ref["force"]).max() < 2e-5             # overwritten, as ever
    ani.close()
