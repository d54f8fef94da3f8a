"""GPU parity tests: libani_hip.so (through the C ABI) against the fp64 CPU oracle and the golden fixtures.

Tolerances (fp32 device arithmetic vs fp64 reference), written here once:
  energy   |dE| <= 1.2e-4 * natoms^0.5 + 3e-4 kcal/mol ... the reference's own fp32 bars are 1.2e-4 kcal/mol on a 30-atom
           system (models/test_models.py:213-214) and 3e-4 Hartree (src/ani_csrc/test_model.cpp:164); we use
           E_TOL = 2e-3 kcal/mol absolute on |E| ~ 5e5..1e7 kcal/mol (4e-9..2e-10 relative), which fp32 network
           outputs summed in fp64 meet.
  forces   F_TOL = 2.3e-3 kcal/mol/A = 1e-4 eV/A (BASELINE.json north_star); typical observed 1e-4.
  virial   V_TOL = 2e-2 kcal/mol absolute on entries of magnitude 1e2..1e4 (sum over ~1e4 pair terms).
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, golden_input, golden_model_path, load_golden
from lammps_ani_amd import harness as hx
from lammps_ani_amd import model_file as mf

pytestmark = pytest.mark.gpu

E_TOL = 2e-3
F_TOL = 2.3e-3
V_TOL = 2e-2


@pytest.fixture(scope="module")
def hip():
    from lammps_ani_amd import ani_hip
    return ani_hip


def _check(got, ref, natoms, label=""):
    de = abs(got["energy"] - ref["energy"])
    df = np.abs(got["force"] - ref["force"]).max()
    dea = np.abs(got["eatom"] - ref["eatom"]).max() if got.get("eatom") is not None and ref.get("eatom") is not None else 0.0
    dv = np.abs(got["virial"] - ref["virial"]).max()
    print(f"{label}: |dE|={de:.2e} max|dF|={df:.2e} max|dEatom|={dea:.2e} max|dV|={dv:.2e}")
    assert de < E_TOL * max(1.0, natoms / 100.0)
    assert df < F_TOL
    assert dea < E_TOL
    assert dv < V_TOL * max(1.0, natoms / 100.0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", ["strict", "compat"])
@pytest.mark.parametrize("half", [False, True], ids=["full", "half"])
def test_hip_matches_golden(case, mode, half, model_cache, hip):
    g = load_golden(case)
    inp = golden_input(g, half=half)
    ani = hip.ANI(golden_model_path(g, model_cache), 0, -1, use_cuaev=(mode == "strict"), use_fullnbr=not half)
    got = ani.compute(inp, ago=0)
    ref = dict(energy=float(g[f"{mode}_energy"]), force=g[f"{mode}_force"], eatom=g[f"{mode}_eatom"], virial=g[f"{mode}_virial"])
    _check(got, ref, inp.nlocal, f"{case}/{mode}/{'half' if half else 'full'}")
    ani.close()


def test_hip_aev_matches_oracle(model_cache, hip):
    """The AEV rows themselves (bucketed order) against the oracle's, 1e-5 absolute on values of O(1)."""
    from oracle import Oracle
    g = load_golden("mixed96_pbc_ani2x_m2")
    inp = golden_input(g)
    p = golden_model_path(g, model_cache)
    ani = hip.ANI(p, 0)
    ani.compute(inp, ago=0)
    v = ani.debug_view()
    A = v.aev_active_length
    assert A == ani.aev_length  # all 7 species occur in this box: nothing to prune
    rows = ani.debug_read(v.d_row_of_centre, inp.nlocal, np.int32)
    aev = ani.debug_read(v.d_aev, (v.nrows, v.aev_stride), np.float32)
    ref = Oracle(p).compute(inp, want_aev=True)
    np.testing.assert_allclose(aev[rows, :A], ref["aev"], rtol=0, atol=1e-5)
    # bucket padding rows stay zero
    mask = np.ones(v.nrows, bool)
    mask[rows] = False
    assert np.all(aev[mask] == 0)
    ani.close()


def test_hip_ago_reuses_list_and_tracks_positions(model_cache, hip):
    """ago > 0: lists are cached on the device (src/ani_csrc/ani.cpp:213-229); new positions must still be used."""
    from oracle import Oracle
    p = model_cache("ani1x", 2, 7)
    s = hx.random_box(80, 4, 10.0, seed=3)
    inp = hx.decompose(s, cutoff=5.2, skin=2.0)
    ani = hip.ANI(p, 0)
    o = Oracle(p)
    ani.compute(inp, ago=0)
    rng = np.random.default_rng(0)
    for step in range(1, 4):
        inp.x = inp.x + rng.normal(0, 0.02, size=inp.x.shape)  # ghosts move independently: fine for a parity check
        got = ani.compute(inp, ago=step)
        _check(got, o.compute(inp), inp.nlocal, f"ago={step}")
    ani.close()


def test_hip_select_models(model_cache, hip):
    from oracle import Oracle
    g = load_golden("water30_pbc_ani2x_m8")
    inp = golden_input(g)
    p = golden_model_path(g, model_cache)
    for n in (1, 4):
        ani = hip.ANI(p, 0, use_num_models=n)
        assert ani.use_num_models == n and ani.num_models == 8
        _check(ani.compute(inp, ago=0), Oracle(p, use_num_models=n).compute(inp), inp.nlocal, f"models={n}")
        ani.close()


def test_hip_water_1500_against_oracle(model_cache, hip):
    """A few-thousand-atom water box (several GEMM tiles per species, ghosts from periodic images)."""
    from oracle import Oracle
    p = model_cache("ani2x", 1, 2024)
    inp = hx.decompose(hx.water_box(1500, seed=9), cutoff=5.1, skin=2.0)
    ani = hip.ANI(p, 0)
    got = ani.compute(inp, ago=0)
    _check(got, Oracle(p).compute(inp), inp.nlocal, "water1500")
    # net force after folding ghosts home is zero to fp32 accumulation noise
    f = got["force"][: inp.nlocal].copy()
    np.add.at(f, inp.owner_lidx, got["force"][inp.nlocal:])
    assert np.abs(f.sum(0)).max() < 5e-2
    ani.close()


@pytest.mark.parametrize("eta_r", [12.0, 30.0, 80.0], ids=["wide", "narrow", "beyond-the-guard"])
def test_hip_radial_widths_other_than_the_published_ones(eta_r, hip, tmp_path):
    """The backward kernel's radial Gaussians come from a recurrence (six exponentials for sixteen shifts) whose factors
    depend on EtaR, the shift spacing and the cutoff; `fast_kind()` sends parameter sets whose factors could leave fp32 to the
    generic kernels.  ANI-2x shaped models with other radial widths, against the oracle: two inside the guard, one beyond
    (reference: the parameters come from the model file, models/lammps_ani.py:130-216 does not fix them)."""
    from oracle import Oracle
    m = mf.synthetic_model("ani2x", 2, 11)
    m.EtaR = eta_r
    p = str(tmp_path / f"ani2x_eta{eta_r:g}.anim")
    mf.write_model(p, m)
    inp = hx.decompose(hx.water_box(900, seed=4), cutoff=5.1, skin=2.0)
    ani = hip.ANI(p, 0)
    got = ani.compute(inp, ago=0)
    _check(got, Oracle(p).compute(inp), inp.nlocal, f"EtaR={eta_r}")
    ani.close()


def test_hip_refuses_cpu(model_cache, hip):
    p = model_cache("tiny", 2, 5)
    with pytest.raises(hip.AniError, match="cpu"):
        hip.ANI(p, -1)
    with pytest.raises(hip.AniError, match="cannot open"):
        hip.ANI("/nonexistent.anim", 0)


def test_hip_empty_and_isolated(model_cache, hip):
    """Edge cases: a rank with zero local atoms; atoms with no neighbours (energy = network(0) + self energy)."""
    from oracle import Oracle
    p = model_cache("tiny", 2, 5)
    ani = hip.ANI(p, 0)
    empty = hx.RankInput(0, 0, np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                         np.zeros(0, np.int32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                         np.zeros(0, np.int32), False)
    got = ani.compute(empty, ago=0)
    assert got["energy"] == 0.0
    s = hx.System(np.array([[0.0, 0, 0], [20.0, 0, 0], [0, 20.0, 0]]), np.array([1, 2, 3], np.int32), np.full(3, -30.0), np.full(3, 30.0), (False,) * 3)
    inp = hx.decompose(s)
    got = ani.compute(inp, ago=0)
    _check(got, Oracle(p).compute(inp), 3, "isolated")
    assert np.all(got["force"] == 0)
    ani.close()


def test_hip_device_api_with_torch_plumbing(model_cache, hip):
    """Device-resident entry point (the Kokkos-overload replacement) driven the way bench.py drives it: torch owns
    the device memory and the stream, forces are ADDED into d_f, ghosts are folded home by comm.GhostExchange."""
    import torch
    from lammps_ani_amd import comm
    from oracle import Oracle
    p = model_cache("ani2x", 2, 2024)
    s = hx.water_box(3000, seed=4)
    inp = hx.decompose(s, cutoff=5.1, skin=2.0)
    dev = torch.device("cuda", 0)
    ani = hip.ANI(p, 0)
    d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
    d_sp = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
    d_il, d_nn, d_jl = (torch.from_numpy(a).to(dev) for a in (inp.ilist, inp.numneigh, inp.jlist))
    d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev)
    d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
    d_ea = torch.zeros(inp.nlocal, dtype=torch.float64, device=dev)
    ex = comm.GhostExchange(inp, s.boxhi - s.boxlo, dev)
    st = torch.cuda.current_stream().cuda_stream
    for ago in range(3):
        d_f.fill_(1.0)  # forces must be accumulated, not overwritten (src/pair_ani_kokkos.cpp:190-191)
        ani.compute_device(inp.ntotal, inp.nlocal, d_sp.data_ptr(), d_x.data_ptr(), inp.npairs, d_il.data_ptr(),
                           d_jl.data_ptr(), d_nn.data_ptr(), ago, d_f.data_ptr(), d_ev.data_ptr(), d_ea.data_ptr(),
                           eflag_atom=True, vflag=True, stream=st)
        f_raw = d_f.view(-1, 3).clone()
        ex.reverse_add(d_f.view(-1, 3))
    torch.cuda.synchronize()
    ref = Oracle(p).compute(inp)
    got = dict(energy=float(d_ev[0]), force=f_raw.cpu().numpy() - 1.0, eatom=d_ea.cpu().numpy(),
               virial=d_ev[1:].cpu().numpy().reshape(3, 3))
    _check(got, ref, inp.nlocal, "device-api")
    folded = ref["force"][: inp.nlocal].copy()
    np.add.at(folded, inp.owner_lidx, ref["force"][inp.nlocal:])
    # after reverse_add: local rows = 1 + own force + (1 + ghost force) per ghost image folded in
    nimg = np.bincount(inp.owner_lidx, minlength=inp.nlocal)[:, None]
    np.testing.assert_allclose(d_f.view(-1, 3)[: inp.nlocal].cpu().numpy() - 1.0 - nimg, folded, rtol=0, atol=F_TOL)
    ani.close()


def test_hip_prunes_columns_of_absent_species(model_cache, hip):
    """Water with the 7-species ANI-2x layout: only the H/O columns (128 of 1008) are computed; the columns that are
    kept equal the oracle's, the dropped ones are exactly zero in the oracle, and energies/forces equal those of the
    full-width run (option prune_absent_species = 0) to fp32 summation-order noise."""
    from oracle import Oracle
    p = model_cache("ani2x", 2, 2024)
    inp = hx.decompose(hx.water_box(600, seed=11), cutoff=5.1, skin=2.0)
    ref = Oracle(p).compute(inp, want_aev=True)
    ani = hip.ANI(p, 0)
    got = ani.compute(inp, ago=0)
    v = ani.debug_view()
    assert v.aev_active_length == 2 * 16 + 3 * 32 == 128
    cm = ani.colmap()
    rows = ani.debug_read(v.d_row_of_centre, inp.nlocal, np.int32)
    aev = ani.debug_read(v.d_aev, (v.nrows, v.aev_stride), np.float32)
    np.testing.assert_allclose(aev[rows][:, : len(cm)], ref["aev"][:, cm], rtol=0, atol=1e-5)
    dropped = np.setdiff1d(np.arange(ani.aev_length), cm)
    assert np.all(ref["aev"][:, dropped] == 0)
    _check(got, ref, inp.nlocal, "pruned")
    ani.set_option("prune_absent_species", 0)
    full = ani.compute(inp, ago=0)
    assert ani.debug_view().aev_active_length == ani.aev_length
    _check(full, ref, inp.nlocal, "full-width")
    assert abs(full["energy"] - got["energy"]) < 1e-3
    assert np.abs(full["force"] - got["force"]).max() < 1e-3
    ani.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", ["strict", "compat"])
@pytest.mark.parametrize("half", [False, True], ids=["full", "half"])
def test_hip_double_precision_matches_golden(case, mode, half, model_cache, hip):
    """precision 'double' (pair_style ... single|double, src/pair_ani.cpp:326-337): fp64 kernels against the fp64
    fixtures at the reference's own fp64 bars — 1e-8 on forces (src/ani_csrc/test_model.cpp:164), 9e-9 relative on the
    energy (yaml epsilon)."""
    g = load_golden(case)
    inp = golden_input(g, half=half)
    ani = hip.ANI(golden_model_path(g, model_cache), 0, -1, use_cuaev=(mode == "strict"), use_fullnbr=not half, use_single=False)
    for ago in (0, 1):
        got = ani.compute(inp, ago=ago)
        assert abs(got["energy"] - float(g[f"{mode}_energy"])) < 9e-9 * abs(float(g[f"{mode}_energy"]))
        np.testing.assert_allclose(got["force"], g[f"{mode}_force"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(got["eatom"], g[f"{mode}_eatom"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(got["virial"], g[f"{mode}_virial"], rtol=0, atol=1e-6)
    ani.close()
