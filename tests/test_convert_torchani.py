"""CPU: the torchani state_dict converter round-trips a model through torchani's key layout and
the result is byte-identical as a model file; bad inputs fail loudly."""
import numpy as np
import pytest
import torch

from lammps_ani_amd import convert_torchani as cv
from lammps_ani_amd import model_file as mf


@pytest.mark.parametrize("kind,nm", [("ani2x", 2), ("ani1x", 3), ("tiny", 2)])
def test_state_dict_round_trip(kind, nm, tmp_path):
    m = mf.synthetic_model(kind, nm, seed=31)
    sd = {("model." + k): torch.as_tensor(v) for k, v in cv.to_state_dict(m).items()}   # torch tensors, with a prefix
    sd["model.aev_computer.triu_index"] = torch.zeros(3, 3)                              # unrelated buffers are ignored
    m2 = cv.from_state_dict(sd, m.species, m.Rcr, m.Rca)
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
    with pytest.raises(KeyError, match="neural_networks"):
        cv.from_state_dict({k: v for k, v in sd.items() if "neural" not in k}, m.species)
