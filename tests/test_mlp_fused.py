"""The fused MLP kernels (a row tile through all six products in one workgroup, activations in registers, weights streamed
through LDS slots) against the per-layer kernels on the same inputs, through the C ABI.  Three forms: the 16-rows-per-wave
kernel (ani_kernels_mlpg.hip, option mlp_fused_gen 1, the default) with 128-row tiles on eight waves and with 64-row tiles on
four, and the 32-rows-per-wave kernel (ani_kernels_mlpf.hip, mlp_fused_gen 0).

Both evaluate BmmEnsemble forward + autograd back to dE/dAEV (models/lammps_ani.py:110,228-230,197) with the same split
arithmetic; they differ in summation order only.  Shapes: water (AEV pruned to 128 columns, one chunk of dE/dAEV tiles),
ANI-1x on four species (384 columns, three chunks), five of the ANI-2x species (560 columns: an odd number of k-steps and a
last narrower chunk), all seven (1008 columns, odd k-steps), one and two ensemble members (mlp_fused = 2 forces the fused kernel for several members), both split
arithmetics.  Bars: the north star's 1e-4 eV/A on forces is 2.3e-3 kcal/mol/A; the two kernels agree 50 times closer.
"""
import numpy as np
import pytest

from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

pytestmark = pytest.mark.gpu

CASES = [("ani2x", 1, "water"), ("ani1x", 1, "mixed4"), ("ani1x", 2, "mixed4"), ("ani2x", 2, "mixed7"), ("ani2x", 1, "mixed5")]


def _box(name):
    if name == "water":
        return hx.water_box(1500, seed=5)
    if name == "mixed4":
        return hx.random_box(64, 4, 9.0, seed=3)
    if name == "mixed5":   # 560 AEV columns: 35 k-steps (odd) and 18 dE/dAEV tiles (a last chunk of two)
        return hx.random_box(500, 5, 20.0, seed=4)
    return hx.random_box(700, 7, 22.0, seed=9)


# (generation, rows per tile, halves): halves 2 = every item of the sixteen-row kernel runs as two half items (the lower half of
# the waves does the rows, the upper half only its share of the loads and the rendezvous); 1 = where the schedule wants them
FORMS = [(1, 128, 1), (1, 64, 1), (0, 0, 0), (1, 128, 2), (1, 64, 2)]


@pytest.mark.parametrize("gen,rows,halves", FORMS, ids=["rows16x8", "rows16x4", "rows32x4", "rows16x8-halves", "rows16x4-halves"])
@pytest.mark.parametrize("mode", [2, 3], ids=["member-items", "members-in-sequence"])
@pytest.mark.parametrize("arith", [1, 2], ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("kind,nm,box", CASES, ids=[f"{k}-m{m}-{b}" for k, m, b in CASES])
def test_fused_mlp_equals_per_layer_kernels(kind, nm, box, arith, mode, gen, rows, halves, tmp_path):
    """mode: mlp_fused 2 = small systems with several members run (tile, member) work items, each member writing its own
    dE/dAEV rows (summed by a second kernel); 3 = a tile's members one after the other in its workgroup."""
    if nm == 1 and mode == 3:
        pytest.skip("one member: both modes are the same kernel path")
    path = str(tmp_path / "m.anim")
    mf.write_model(path, mf.synthetic_model(kind, nm, seed=2024))
    inp = hx.decompose(_box(box))
    out = {}
    for fused in (0, mode):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused", fused)
        ani.set_option("mlp_arith", arith)
        ani.set_option("mlp_fused_gen", gen)
        ani.set_option("mlp_fused_rows", rows)
        ani.set_option("mlp_fused_halves", halves)
        out[fused] = ani.compute(inp, ago=0)
        if fused:
            want = "mlp_fused<" if gen == 0 else ("mlp_fused16<%d, %d>" % (3 if arith == 1 else 2, 8 if rows == 128 else 4))
            assert ani.last_mlp_kernel().startswith(want), ani.last_mlp_kernel()   # a second step on the cached list: the tile counter and the ring start over
            again = ani.compute(inp, ago=1)
            assert np.array_equal(again["force"], out[fused]["force"]) or np.abs(again["force"] - out[fused]["force"]).max() < 1e-4
        ani.close()
    assert np.isfinite(out[mode]["energy"])
    assert abs(out[mode]["energy"] - out[0]["energy"]) < 2e-3
    # the exact split agrees to the order of the fp32 sums; the two-term fp16 split carries 2^-22 per operand, and the
    # 16x16x32 instruction sums 32 products where the per-layer kernels sum 16
    assert np.abs(out[mode]["force"] - out[0]["force"]).max() < (2e-4 if arith == 1 else 5e-4)
    assert np.abs(out[mode]["eatom"] - out[0]["eatom"]).max() < 1e-4
    assert np.abs(out[mode]["virial"] - out[0]["virial"]).max() < 2e-2


def test_default_choice_of_mlp_kernels(tmp_path):
    """mlp_fused = 1 (default): eight members run the fused kernel's (tile, member) work items (= mlp_fused 2, bit for bit:
    the same kernels in the same order); with the 32-row generation one member on a small box runs the chained per-layer
    launch (= mlp_fused 0), with the 16-row generation its 64-row form."""
    p8, p1 = str(tmp_path / "m8.anim"), str(tmp_path / "m1.anim")
    mf.write_model(p8, mf.synthetic_model("ani2x", 8, seed=7))
    mf.write_model(p1, mf.synthetic_model("ani2x", 1, seed=7))
    inp = hx.decompose(hx.water_box(600, seed=2))

    def run(path, fused, gen=None):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused", fused)
        if gen is not None:
            ani.set_option("mlp_fused_gen", gen)
        out = ani.compute(inp, ago=0)
        out["kernel"] = ani.last_mlp_kernel()
        ani.close()
        return out

    # per-atom energies bit for bit; the total is a sum of block sums added with atomics in arrival order: an ulp or two
    a, b = run(p8, 1), run(p8, 2)
    assert abs(a["energy"] - b["energy"]) <= 4e-16 * abs(b["energy"]) * 3 and np.array_equal(a["eatom"], b["eatom"])
    a, b = run(p1, 1, gen=0), run(p1, 0, gen=0)
    assert abs(a["energy"] - b["energy"]) <= 4e-16 * abs(b["energy"]) * 3 and np.array_equal(a["eatom"], b["eatom"]) and a["kernel"] == "mlp_chain"
    a = run(p1, 1)
    assert a["kernel"] == "mlp_fused16<3, 4>" and abs(a["energy"] - b["energy"]) < 2e-3
