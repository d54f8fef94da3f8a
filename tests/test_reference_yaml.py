"""The reference's OWN golden vectors for this path: tests/lammps-unittest/*/*.yaml (LAMMPS test_pair_style files) hold
the energy, forces and pair virial of the trained ANI-2x ensemble on tests/water-0.8nm.data, open and periodic, before
and after 4 NVE steps of 0.1 fs, for {cuaev,pyaev} x {full,half} x {single,double}.  They are committed as data in
tests/golden/reference_yaml/reference_yaml.json (imported by the script next to it).

They pin nothing until the trained weights exist in this build's format: set ANI2X_MODEL=/path/ani2x.anim (made by
lammps_ani_amd.convert_torchani where torchani is installed) and the GPU test below runs every fixture through the C
ABI with the acceptance rule of LAMMPS' test_pair_style (relative error <= epsilon of the yaml file).  Without the
variable it is skipped, and only the CPU checks of the fixtures themselves run."""
import json
import os

import numpy as np
import pytest

from lammps_ani_amd import harness as hx

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "reference_yaml", "reference_yaml.json")))
DATA = os.path.join(HERE, "golden", "water-0.8nm.data")
MODEL = os.environ.get("ANI2X_MODEL", "")
FTM2V = 1.0 / 48.88821291 / 48.88821291   # LAMMPS `units real`
MASS = {1: 1.008, 2: 12.0107, 3: 14.0067, 4: 15.999, 5: 32.06, 6: 18.998403163, 7: 35.45}


def test_fixtures_are_well_formed_and_mutually_consistent():
    assert len(FIX) == 15
    by = {(f["suite"], f["name"]): f for f in FIX}
    for f in FIX:
        assert f["natoms"] == 30 and np.asarray(f["init_forces"]).shape == (30, 3) and np.asarray(f["run_forces"]).shape == (30, 3)
        tol = 2e-8 if f["precision"] == "double" else 2e-2
        assert np.abs(np.sum(f["init_forces"], 0)).max() < tol   # forces of a closed system sum to zero
        assert (f["init_stress"] is None) == (f["aev"] == "cuaev")   # the reference's cuaev path returns no virial
    # the same physics through different code paths of the reference: fp64 cpu == fp64 cuda; fp32 close to fp64
    d = "test_ani2x_nocuaev_double_half"
    for a, b in (("manybody-pair-ani-double-cpu", "manybody-pair-ani-double-cuda"),
                 ("manybody-pair-ani-pbc-double-cpu", "manybody-pair-ani-pbc-double-cuda")):
        assert by[(d, a)]["init_vdwl"] == by[(d, b)]["init_vdwl"]
        assert np.abs(np.asarray(by[(d, a)]["init_forces"]) - np.asarray(by[(d, b)]["init_forces"])).max() < 1e-9
    s = by[("test_ani2x_nocuaev_single_half", "manybody-pair-ani-single-cpu")]
    assert np.abs(np.asarray(s["init_forces"]) - np.asarray(by[(d, "manybody-pair-ani-double-cpu")]["init_forces"])).max() < 1e-3
    # the water molecules of the data file: what the fixtures' 30 atoms are
    sysm = hx.read_lammps_data(DATA)
    assert len(sysm.x) == 30 and sorted(set(sysm.types.tolist())) == [1, 4]


def _close(val, ref, eps):
    """EXPECT_FP_LE_WITH_EPS of LAMMPS' unittest/force-styles: relative where the magnitude exceeds eps."""
    val, ref = np.asarray(val, float), np.asarray(ref, float)
    err = np.abs(val - ref)
    den = np.maximum(np.abs(val), np.abs(ref))
    err = np.where(den > eps, err / np.maximum(den, 1e-300), err)
    return float(err.max())


def _evaluate(ani, sysm, fx):
    inp = hx.decompose(sysm, cutoff=fx["cutoff"], skin=2.0, half=(fx["nbr"] == "half"))
    out = ani.compute(inp, ago=0)
    f = out["force"][: inp.nlocal].copy()
    np.add.at(f, inp.owner_lidx, out["force"][inp.nlocal:])
    v = out["virial"]
    return out["energy"], f, [v[0, 0], v[1, 1], v[2, 2], v[0, 1], v[0, 2], v[1, 2]]


@pytest.mark.gpu
@pytest.mark.skipif(not MODEL, reason="needs the trained ANI-2x ensemble in this build's format: ANI2X_MODEL=/path/ani2x.anim")
@pytest.mark.parametrize("fx", FIX, ids=[f"{f['suite']}/{f['name']}" for f in FIX])
def test_reference_golden_vectors(fx):
    from lammps_ani_amd import ani_hip
    base = hx.read_lammps_data(DATA)
    sysm = hx.System(base.x.copy(), base.types, base.boxlo, base.boxhi, periodic=(fx["periodic"],) * 3)
    ani = ani_hip.ANI(MODEL, 0, fx["num_models"], use_cuaev=(fx["aev"] == "cuaev"), use_fullnbr=(fx["nbr"] == "full"),
                      use_single=(fx["precision"] == "single"))
    eps = fx["epsilon"]
    e, f, v = _evaluate(ani, sysm, fx)
    assert _close(e, fx["init_vdwl"], eps) <= eps
    assert _close(f, fx["init_forces"], eps) <= eps
    if fx["init_stress"] is not None:
        assert _close(v, fx["init_stress"], 10 * eps) <= 10 * eps
    # `run 4` with fix nve, timestep 0.1 fs, velocities zero (in.ani; neigh_modify every 2 delay 0 check no)
    m = np.array([MASS[t] for t in sysm.types])[:, None]
    vel = np.zeros_like(sysm.x)
    dt = fx["timestep_fs"]
    for _ in range(fx["run_steps"]):
        vel += 0.5 * dt * FTM2V * f / m
        sysm.x += dt * vel
        e, f, v = _evaluate(ani, sysm, fx)
        vel += 0.5 * dt * FTM2V * f / m
    assert _close(e, fx["run_vdwl"], eps) <= eps
    assert _close(f, fx["run_forces"], eps) <= eps * 10
    if fx["run_stress"] is not None:
        assert _close(v, fx["run_stress"], 10 * eps) <= 10 * eps
    ani.close()
