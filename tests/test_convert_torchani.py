"""CPU: the torchani state_dict converter.  The reference imports the torchani 2.x API
(models/ani_models.py:5-7,17; models/lammps_ani.py:6) whose module names are not in the reference tree, so the converter
discovers the networks structurally; here the same synthetic model is spelled three ways — the torchani <= 2.2 keys,
a members / atomics / layers / final_layer + radial / angular term spelling, and a single un-ensembled model under an
extra prefix — and every spelling must give a byte-identical model file.  Bad inputs fail loudly and list the keys."""
import numpy as np
import pytest
import torch

from lammps_ani_amd import convert_torchani as cv
from lammps_ani_amd import model_file as mf


@pytest.mark.parametrize("layout", ["legacy", "terms"])
@pytest.mark.parametrize("kind,nm", [("ani2x", 2), ("ani1x", 3), ("tiny", 2)])
def test_state_dict_round_trip(kind, nm, layout, tmp_path):
    m = mf.synthetic_model(kind, nm, seed=31)
    sd = {("model." + k): torch.as_tensor(v) for k, v in cv.to_state_dict(m, layout).items()}   # torch tensors, with a prefix
    sd["model.aev_computer.triu_index"] = torch.zeros(3, 3)                              # unrelated buffers are ignored
    sd["model.neural_networks.some_table.weight"] = torch.zeros(4, 4)                    # a 2-D weight that names no species
    # the "terms" spelling carries its cutoffs as buffers: they must win over (deliberately wrong) arguments
    m2 = cv.from_state_dict(sd, m.species, m.Rcr, m.Rca) if layout == "legacy" else cv.from_state_dict(sd, m.species, 9.9, 9.9)
    a, b = str(tmp_path / "a.anim"), str(tmp_path / "b.anim")
    mf.write_model(a, m)
    mf.write_model(b, m2)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_repulsion_tables_round_trip(tmp_path):
    m = mf.synthetic_model("ani1x", 1, seed=2, repulsion=True)
    m2 = cv.from_state_dict(cv.to_state_dict(m), m.species, m.Rcr, m.Rca, rep_cutoff=m.repulsion["cutoff"])
    a, b = str(tmp_path / "a.anim"), str(tmp_path / "b.anim")
    mf.write_model(a, m)
    mf.write_model(b, m2)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert cv.from_state_dict(cv.to_state_dict(m), m.species, m.Rcr, m.Rca).repulsion is None   # only on request


def test_cli_and_single_network_layout(tmp_path):
    m = mf.synthetic_model("tiny", 1, seed=4)
    sd = {k.replace("neural_networks.0.", "neural_networks."): torch.as_tensor(v) for k, v in cv.to_state_dict(m).items()}
    p, out = str(tmp_path / "sd.pt"), str(tmp_path / "m.anim")
    torch.save(sd, p)
    cv.main([p, out, "--species", *m.species])
    r = mf.read_model(out)
    assert r.num_models == 1 and r.dims == m.dims
    assert all(np.array_equal(r.weights[0][s][l][0], m.weights[0][s][l][0]) for s in range(3) for l in range(4))


def test_bad_inputs_fail_loudly():
    m = mf.synthetic_model("tiny", 1, seed=4)
    sd = cv.to_state_dict(m)
    with pytest.raises(KeyError, match="species"):
        cv.from_state_dict(sd, ["H", "C", "N"])
    with pytest.raises(ValueError, match="AEV length"):
        cv.from_state_dict(sd, ["H", "C"] + ["O"] * 0 + ["O"], 5.1, 3.5) if False else cv.from_state_dict(
            {**sd, "aev_computer.ShfR": np.arange(7.0)}, m.species)
    with pytest.raises(KeyError, match="no atomic networks found") as ei:
        cv.from_state_dict({k: v for k, v in sd.items() if "neural" not in k}, m.species)
    assert "aev_computer.ShfR  (1, 5)" in str(ei.value) and "energy_shifter.self_energies  (3,)" in str(ei.value)   # every key, with shape
    with pytest.raises(KeyError, match="AEV constants \\['Zeta'\\]"):
        cv.from_state_dict({k: v for k, v in sd.items() if not k.endswith("Zeta")}, m.species)
    broken = dict(sd)
    broken["neural_networks.0.H.2.weight"] = np.zeros((5, 7), np.float32)   # does not chain onto layer 0
    broken["neural_networks.0.H.2.bias"] = np.zeros(5, np.float32)
    with pytest.raises(ValueError, match="do not chain"):
        cv.from_state_dict(broken, m.species)


def test_final_layer_is_found_by_shape_and_members_may_be_unordered_in_the_dict():
    """Layer order must not depend on dictionary order or on a particular index step; a layer with one output row is the
    last one whatever it is called."""
    m = mf.synthetic_model("tiny", 2, seed=9)
    sd = cv.to_state_dict(m, "terms")
    renamed = {}
    for k, v in reversed(list(sd.items())):
        k = k.replace("final_layer", "layers.99").replace("layers.", "seq.")
        renamed["wrapper.potential." + k] = v
    m2 = cv.from_state_dict(renamed, m.species)
    assert m2.dims == m.dims and m2.num_models == 2
    for mi in range(2):
        for s in range(len(m.species)):
            for l in range(len(m.dims[0]) - 1):
                assert np.array_equal(m2.weights[mi][s][l][0], m.weights[mi][s][l][0])
