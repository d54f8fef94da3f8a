"""The one reference-held vector that needs no trained weights: the 411-pair half neighbour list embedded in the
reference's C++ known-answer test (src/ani_csrc/test_model.cpp:84-120; fixture tests/golden/reference_kat/, data only).

It is what the reference's LAMMPS produced for the 30-atom water cluster at cutoff 5.1 + skin 2.0 = 7.1 A with open
boundaries, so it pins — bit-exactly, as pair SETS (integer work) — every neighbour-list builder of this repository:

  * the host harness (the LAMMPS stand-in that feeds all other tests): half and full list            [CPU]
  * ani_build_list (host arrays) and ani_build_list_device, through the C ABI                         [-m gpu]
  * the half -> per-centre expansion inside ani_compute_half: the forces from the reference's list equal the
    forces from our own list of the same system, bit for bit                                          [-m gpu]

Also checked: the fixture's own consistency (the expected forces of the trained model sum to zero; every listed pair
is inside 7.1 A, every unlisted one outside).  The expected energy / forces themselves need the trained ANI-2x
ensemble and are exercised by test_reference_kat_trained_model only when ANI2X_MODEL points at a
converted model file.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from lammps_ani_amd import harness as hx

KAT = json.load(open(os.path.join(GOLDEN, "reference_kat", "test_model_kat.json")))
CUT = KAT["cutoff"] + KAT["skin"]


def kat_system():
    x = np.array(KAT["coords"], dtype=np.float64).reshape(-1, 3)
    types = np.array(KAT["species"], dtype=np.int32) + 1        # LAMMPS type = species + 1 (src/pair_ani.cpp:110)
    lo, hi = x.min(0) - 1.0, x.max(0) + 1.0
    return hx.System(x, types, lo, hi, periodic=(False, False, False))


def kat_pairs():
    a = np.array(KAT["atom_index12"], dtype=np.int64)
    n = len(a) // 2
    return {(int(min(i, j)), int(max(i, j))) for i, j in zip(a[:n], a[n:])}


def pairs_of(numneigh, jlist, ilist=None):
    off = np.concatenate([[0], np.cumsum(numneigh)])
    out = []
    for ii in range(len(numneigh)):
        i = ii if ilist is None else int(ilist[ii])
        out += [(i, int(j)) for j in jlist[off[ii]:off[ii + 1]]]
    return out


def test_fixture_is_self_consistent():
    x = np.array(KAT["coords"]).reshape(-1, 3)
    ref = kat_pairs()
    assert len(ref) == 411 == len(KAT["atom_index12"]) // 2      # no duplicate pairs
    d = np.linalg.norm(x[:, None] - x[None], axis=-1)
    listed = np.array([d[i, j] for i, j in ref])
    unlisted = np.array([d[i, j] for i in range(30) for j in range(i + 1, 30) if (i, j) not in ref])
    assert listed.max() <= CUT < unlisted.min()
    assert len(unlisted) == 435 - 411
    # Newton's third law on the trained model's expected forces (open boundary, all atoms local)
    f = np.array(KAT["expected_force_kcal_mol_A"]).reshape(-1, 3)
    assert np.abs(f.sum(0)).max() < 1e-10
    # the coordinates are those of the reference's data file (tests/water-0.8nm.data, committed as a fixture)
    data = hx.read_lammps_data(os.path.join(GOLDEN, "water-0.8nm.data"))
    assert np.allclose(data.x, x, atol=5e-5) and np.array_equal(data.types - 1, KAT["species"])


def test_host_harness_reproduces_reference_half_list():
    inp = hx.decompose(kat_system(), cutoff=KAT["cutoff"], skin=KAT["skin"], half=True)
    assert inp.nghost == 0 and inp.nlocal == 30
    got = pairs_of(inp.numneigh, inp.jlist)
    assert len(got) == 411                                         # each pair once
    assert {(min(i, j), max(i, j)) for i, j in got} == kat_pairs()


def test_host_harness_full_list_is_the_symmetric_closure():
    inp = hx.decompose(kat_system(), cutoff=KAT["cutoff"], skin=KAT["skin"], half=False)
    got = pairs_of(inp.numneigh, inp.jlist)
    ref = kat_pairs()
    assert len(got) == 822 and set(got) == {(i, j) for i, j in ref} | {(j, i) for i, j in ref}


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip():
    from lammps_ani_amd import ani_hip
    return ani_hip


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["ani_build_list", "ani_build_list_device"])
def test_device_builders_reproduce_reference_list(entry, model_cache, hip):
    sysm = kat_system()
    ref = kat_pairs()
    full = {(i, j) for i, j in ref} | {(j, i) for i, j in ref}
    ani = hip.ANI(model_cache("ani2x", 1, 11), 0)
    species = (sysm.types - 1).astype(np.int64)
    if entry == "ani_build_list":
        n = ani.build_list(species, sysm.x, 30, CUT)
    else:
        import torch
        dev = torch.device("cuda:0")
        x = torch.as_tensor(sysm.x, dtype=torch.float64, device=dev).contiguous()
        sp = torch.as_tensor(species.astype(np.int32), device=dev)
        n = ani.build_list_device(30, 30, sp.data_ptr(), x.data_ptr(), CUT, sysm.x.min(0) - 0.25, sysm.x.max(0) + 0.25)
        torch.cuda.synchronize()
    assert n == 822
    nn, jl = ani.debug_list(30)
    got = pairs_of(nn, jl)
    assert len(got) == 822 and set(got) == full
    ani.close()


@pytest.mark.gpu
def test_reference_half_list_through_compute_half_equals_own_list(model_cache, hip):
    """ani_compute_half fed the reference's own atom_index12 (its pair order, its i/j orientation) gives the forces of
    the list our harness builds for the same cluster: same neighbour sets per centre after the expansion, so the
    kernels see species-sorted segments that differ at most in order inside a species group — fp32 sums of <= 29
    terms, compared at the fp32 force bar; in precision double at 1e-9."""
    sysm = kat_system()
    mine = hx.decompose(sysm, cutoff=KAT["cutoff"], skin=KAT["skin"], half=True)
    a12 = np.array(KAT["atom_index12"], dtype=np.int64)
    n = len(a12) // 2
    # a RankInput whose half list is the reference's: ilist order by first index, j = second index
    order = np.argsort(a12[:n], kind="stable")
    theirs = hx.RankInput(nlocal=30, nghost=0, x=mine.x, types=mine.types, tag=mine.tag, owner_rank=mine.owner_rank,
                          owner_lidx=mine.owner_lidx, shift=mine.shift, ilist=np.arange(30, dtype=np.int32),
                          numneigh=np.bincount(a12[:n], minlength=30).astype(np.int32),
                          jlist=a12[n:][order].astype(np.int32), half=True)
    assert np.array_equal(theirs.atom_index12()[:n], a12[:n][order])
    for single, tol in ((True, 2.3e-3), (False, 1e-9)):
        ani = hip.ANI(model_cache("ani2x", 8, 2024), 0, -1, use_cuaev=False, use_fullnbr=False, use_single=single)
        a = ani.compute(mine, ago=0)
        b = ani.compute(theirs, ago=0)
        assert np.abs(a["force"] - b["force"]).max() < tol
        assert abs(a["energy"] - b["energy"]) < (2e-3 if single else 1e-8)
        assert np.abs(a["force"].sum(0)).max() < (5e-3 if single else 1e-8)
        ani.close()


@pytest.mark.gpu
def test_reference_kat_trained_model(hip):
    """src/ani_csrc/test_model.cpp:121-171 verbatim (pyaev, half list, all 8 members; thresholds 1e-8 fp64 / 3e-4 fp32 on
    Hartree energy and kcal/mol/A forces) — runs only with the trained ANI-2x ensemble converted by convert_torchani.py."""
    path = os.environ.get("ANI2X_MODEL")
    if not path:
        pytest.skip("trained ANI-2x weights are not in the container (SURVEY.md 8c): set ANI2X_MODEL=<converted model file>")
    sysm = kat_system()
    inp = hx.decompose(sysm, cutoff=KAT["cutoff"], skin=KAT["skin"], half=True)
    f_ref = np.array(KAT["expected_force_kcal_mol_A"]).reshape(-1, 3)
    for single in (False, True):
        ani = hip.ANI(path, 0, -1, use_cuaev=False, use_fullnbr=False, use_single=single)
        out = ani.compute(inp, ago=0)
        thr = KAT["threshold_fp32"] if single else KAT["threshold_fp64"]
        assert abs(out["energy"] / 627.5094738898777 - KAT["expected_energy_hartree"]) < thr
        assert np.abs(out["force"] - f_ref).max() < thr
        ani.close()
