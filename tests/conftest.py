import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load()
from lammps_ani_amd import harness as hx  # noqa: E402
from lammps_ani_amd import model_file as mf  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["water30_pbc_ani2x_m8", "water30_open_ani2x_m8", "mixed64_pbc_ani1x_m2", "mixed40_pbc_tiny_m3",
                "mixed96_pbc_ani2x_m2", "mixed64_pbc_ani1x_m2_rep"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_cache(tmp_path_factory):
    """(kind, num_models, seed) -> path of a freshly generated model file (weights are never committed)."""
    d = tmp_path_factory.mktemp("models")
    cache = {}

    def get(kind, num_models, seed, repulsion=False):
        key = (kind, int(num_models), int(seed), bool(repulsion))
        if key not in cache:
            p = str(d / f"{kind}_m{num_models}_s{seed}{'_rep' if repulsion else ''}.anim")
            mf.write_model(p, mf.synthetic_model(kind, int(num_models), int(seed), repulsion=bool(repulsion)))
            cache[key] = p
        return cache[key]

    return get


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_input(g, half=False):
    """Rebuild the harness.RankInput the fixture was generated with (lists are stored, not rebuilt)."""
    nlocal = int(g["nlocal"])
    x = g["x"]
    ng = x.shape[0] - nlocal
    return hx.RankInput(
        nlocal=nlocal, nghost=ng, x=x, types=g["types"], tag=np.zeros(x.shape[0], np.int32),
        owner_rank=np.zeros(ng, np.int32), owner_lidx=g["owner_lidx"], shift=np.zeros((ng, 3), np.int32),
        ilist=np.arange(nlocal, dtype=np.int32),
        numneigh=g["half_numneigh"] if half else g["numneigh"],
        jlist=g["half_jlist"] if half else g["jlist"], half=half)


def golden_model_path(g, model_cache):
    p = model_cache(str(g["kind"]), int(g["num_models"]), int(g["seed"]), bool(int(g["repulsion"])) if "repulsion" in g else False)
    with open(p, "rb") as f:
        crc = zlib.crc32(f.read())
    assert crc == int(g["model_crc"]), "synthetic model generator drifted from the one the fixtures were made with"
    return p
