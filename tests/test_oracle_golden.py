"""CPU tests: the oracle (oracle/ani_oracle.c) against the committed golden fixtures and against itself.

The fixtures come from tests/golden/make_golden.py (independent torch-autograd restatement, fp64).
Tolerances: both sides are fp64, so agreement is demanded to 1e-9 relative on energies (|E| ~ 5e5 kcal/mol
=> 5e-4 abs would be far too loose; we use 2e-7 kcal/mol abs) and 1e-8 kcal/mol/A on forces — the reference's
own fp64 thresholds are 1e-8 (src/ani_csrc/test_model.cpp:164) and 9e-9 relative (yaml epsilon).
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, golden_input, golden_model_path, load_golden
from lammps_ani_amd import harness as hx
from lammps_ani_amd import model_file as mf
from oracle import Oracle


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("mode", ["strict", "compat"])
@pytest.mark.parametrize("half", [False, True], ids=["full", "half"])
def test_oracle_matches_golden(case, mode, half, model_cache):
    g = load_golden(case)
    inp = golden_input(g, half=half)
    o = Oracle(golden_model_path(g, model_cache))
    r = o.compute(inp, radial_compat=(mode == "compat"), want_aev=True)
    assert abs(r["energy"] - float(g[f"{mode}_energy"])) < 2e-7
    np.testing.assert_allclose(r["force"], g[f"{mode}_force"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(r["eatom"], g[f"{mode}_eatom"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(r["virial"], g[f"{mode}_virial"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(r["aev"], g[f"{mode}_aev"], rtol=0, atol=1e-12)


def test_oracle_fp32_close_to_fp64(model_cache):
    """fp32 restatement stays within the reference's fp32 thresholds (3e-4, src/ani_csrc/test_model.cpp:164)."""
    g = load_golden("water30_pbc_ani2x_m8")
    inp = golden_input(g)
    p = golden_model_path(g, model_cache)
    r64 = Oracle(p).compute(inp)
    r32 = Oracle(p, fp32=True).compute(inp)
    assert abs(r64["energy"] - r32["energy"]) / 627.5094738898777 < 3e-4
    assert np.abs(r64["force"] - r32["force"]).max() < 3e-4


def test_select_models_first_n(model_cache):
    """use_num_models=n takes the first n members (models/lammps_ani.py:342)."""
    g = load_golden("mixed40_pbc_tiny_m3")
    inp = golden_input(g)
    p3 = golden_model_path(g, model_cache)
    m = mf.read_model(p3).select_models(2)
    p2 = p3.replace(".anim", "_first2.anim")
    mf.write_model(p2, m)
    a = Oracle(p3, use_num_models=2).compute(inp)
    b = Oracle(p2).compute(inp)
    assert abs(a["energy"] - b["energy"]) < 1e-9
    np.testing.assert_allclose(a["force"], b["force"], rtol=0, atol=1e-11)  # omp atomics: order-dependent last bits


def test_eatom_sums_to_total_and_net_force_zero(model_cache):
    """atomic=True sum == total (models/test_models.py:226-229); folded ghost forces sum to zero."""
    g = load_golden("mixed64_pbc_ani1x_m2")
    inp = golden_input(g)
    r = Oracle(golden_model_path(g, model_cache)).compute(inp)
    assert abs(r["eatom"].sum() - r["energy"]) < 1e-6
    f = r["force"][: inp.nlocal].copy()
    np.add.at(f, inp.owner_lidx, r["force"][inp.nlocal:])
    assert np.abs(f.sum(0)).max() < 1e-9


def test_virial_matches_finite_strain(model_cache):
    """virial == -dE/d(strain) for a homogeneous deformation of box and atoms (independent of make_golden)."""
    p = model_cache("tiny", 2, 5)
    s = hx.random_box(36, 3, 8.0, seed=2)
    o = Oracle(p)

    def energy(eps):
        F = np.eye(3) + eps
        # orthogonal box only: use diagonal strains
        s2 = hx.System(s.x @ F.T, s.types, s.boxlo * np.diag(F), s.boxhi * np.diag(F), s.periodic)
        return o.compute(hx.decompose(s2))["energy"]

    r = o.compute(hx.decompose(s))
    h = 1e-6
    for k in range(3):
        e = np.zeros((3, 3))
        e[k, k] = h
        fd = -(energy(e) - energy(-e)) / (2 * h)
        assert abs(fd - r["virial"][k, k]) < 2e-3 * max(1.0, abs(fd)), (k, fd, r["virial"][k, k])


def test_rank_count_invariance(model_cache):
    """1 vs 2x1x1 vs 2x2x2 bricks give the same folded forces and total energy."""
    p = model_cache("tiny", 2, 5)
    s = hx.random_box(60, 3, 16.0, seed=4, min_dist=1.2)
    o = Oracle(p)

    def run(grid):
        P = grid[0] * grid[1] * grid[2]
        F = np.zeros((s.natoms, 3))
        E = 0.0
        ins = [hx.decompose(s, grid, r) for r in range(P)]
        outs = [o.compute(i) for i in ins]
        for inp, r in zip(ins, outs):
            E += r["energy"]
            np.add.at(F, inp.tag[: inp.nlocal], r["force"][: inp.nlocal])
            # ghost forces go home to the owner's global atom
            np.add.at(F, inp.tag[inp.nlocal:], r["force"][inp.nlocal:])
        return E, F

    E1, F1 = run((1, 1, 1))
    for grid in [(2, 1, 1), (2, 2, 2)]:
        E, F = run(grid)
        assert abs(E - E1) < 1e-6
        np.testing.assert_allclose(F, F1, rtol=0, atol=1e-9)
