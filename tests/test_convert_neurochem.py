"""CPU: the NeuroChem-directory converter round-trips a model through the file layout restated in
lammps_ani_amd/convert_neurochem.py (plain-text and bz2 network descriptions) to a byte-identical model file; bad inputs
fail loudly.  No real NeuroChem model is in the container (external/ani-1xnr is an empty submodule of the reference)."""
import os

import numpy as np
import pytest

from lammps_ani_amd import convert_neurochem as cn
from lammps_ani_amd import model_file as mf


@pytest.mark.parametrize("kind,nm,compressed", [("ani1x", 2, False), ("ani1x", 1, True), ("tiny", 3, False)])
def test_neurochem_round_trip(kind, nm, compressed, tmp_path):
    m = mf.synthetic_model(kind, nm, seed=17)
    info = cn.write_neurochem(m, str(tmp_path / "nc"), compressed=compressed)
    out = str(tmp_path / "b.anim")
    cn.main([info, out])
    ref = str(tmp_path / "a.anim")
    mf.write_model(ref, m)
    assert open(ref, "rb").read() == open(out, "rb").read()


def test_neurochem_bad_inputs_fail_loudly(tmp_path):
    m = mf.synthetic_model("tiny", 1, seed=3)
    root = str(tmp_path / "nc")
    info = cn.write_neurochem(m, root)
    nets = os.path.join(root, "train0", "networks")
    # truncated weight file
    w = os.path.join(nets, f"ANN-{m.species[0]}-l1.wparam")
    data = open(w, "rb").read()
    open(w, "wb").write(data[:-4])
    with pytest.raises(ValueError, match="holds"):
        cn.from_info_file(info)
    open(w, "wb").write(data)
    # an activation this build does not evaluate
    nnf = os.path.join(nets, f"ANN-{m.species[0]}.nnf")
    text = open(nnf).read()
    open(nnf, "w").write(text.replace("activation=9", "activation=5", 1))
    with pytest.raises(ValueError, match="CELU"):
        cn.from_info_file(info)
    open(nnf, "w").write(text)
    # a species without self energy
    sae = os.path.join(root, "sae.dat")
    lines = open(sae).read().splitlines()
    open(sae, "w").write("\n".join(lines[1:]) + "\n")
    with pytest.raises(KeyError, match="self energy"):
        cn.from_info_file(info)
    open(sae, "w").write("\n".join(lines) + "\n")
    assert cn.from_info_file(info).dims == m.dims
    assert np.array_equal(cn.from_info_file(info).weights[0][1][2][0], m.weights[0][1][2][0])
