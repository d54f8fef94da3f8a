"""The fused MLP kernel (ani_kernels_mlpf.hip: a 128-row tile through all six products in one workgroup, activations in
registers, weights streamed through an LDS ring) against the per-layer kernels on the same inputs, through the C ABI.

Both evaluate BmmEnsemble forward + autograd back to dE/dAEV (models/lammps_ani.py:110,228-230,197) with the same split
arithmetic; they differ in summation order only.  Shapes: water (AEV pruned to 128 columns, one chunk of dE/dAEV tiles),
ANI-1x on four species (384 columns, three chunks), all seven ANI-2x species (1008 columns, an odd number of k-steps and a
last narrower chunk), one and two ensemble members (mlp_fused = 2 forces the fused kernel for several members), both split
arithmetics.  Bars: the north star's 1e-4 eV/A on forces is 2.3e-3 kcal/mol/A; the two kernels agree 50 times closer.
"""
import numpy as np
import pytest

from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

pytestmark = pytest.mark.gpu

CASES = [("ani2x", 1, "water"), ("ani1x", 1, "mixed4"), ("ani1x", 2, "mixed4"), ("ani2x", 2, "mixed7")]


def _box(name):
    if name == "water":
        return hx.water_box(1500, seed=5)
    if name == "mixed4":
        return hx.random_box(64, 4, 9.0, seed=3)
    return hx.random_box(700, 7, 22.0, seed=9)


@pytest.mark.parametrize("arith", [1, 2], ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("kind,nm,box", CASES, ids=[f"{k}-m{m}-{b}" for k, m, b in CASES])
def test_fused_mlp_equals_per_layer_kernels(kind, nm, box, arith, tmp_path):
    path = str(tmp_path / "m.anim")
    mf.write_model(path, mf.synthetic_model(kind, nm, seed=2024))
    inp = hx.decompose(_box(box))
    out = {}
    for fused in (0, 2):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused", fused)
        ani.set_option("mlp_arith", arith)
        out[fused] = ani.compute(inp, ago=0)
        if fused:   # a second step on the cached list: the tile counter and the ring start over
            again = ani.compute(inp, ago=1)
            assert np.array_equal(again["force"], out[fused]["force"]) or np.abs(again["force"] - out[fused]["force"]).max() < 1e-4
        ani.close()
    assert np.isfinite(out[2]["energy"])
    assert abs(out[2]["energy"] - out[0]["energy"]) < 2e-3
    assert np.abs(out[2]["force"] - out[0]["force"]).max() < 2e-4
    assert np.abs(out[2]["eatom"] - out[0]["eatom"]).max() < 1e-4
    assert np.abs(out[2]["virial"] - out[0]["virial"]).max() < 2e-2


def test_several_members_take_the_per_layer_kernels_by_default(tmp_path):
    """mlp_fused = 1 (default) with 8 members must give what mlp_fused = 0 gives bit for bit: the same kernels ran."""
    path = str(tmp_path / "m8.anim")
    mf.write_model(path, mf.synthetic_model("ani2x", 8, seed=7))
    inp = hx.decompose(hx.water_box(600, seed=2))
    res = []
    for fused in (0, 1):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused", fused)
        res.append(ani.compute(inp, ago=0))
        ani.close()
    assert res[0]["energy"] == res[1]["energy"]
    assert np.array_equal(res[0]["eatom"], res[1]["eatom"])
