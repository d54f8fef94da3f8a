"""Imports the golden vectors of the reference's own LAMMPS unit tests (tests/lammps-unittest/*/*.yaml: expected
energy, forces and virial of the TRAINED ANI-2x ensemble on tests/water-0.8nm.data, before and after 4 NVE steps) into
one JSON fixture.  Data only: the numbers and the pair_style arguments, nothing else of the files.

Run where /root/reference exists:  python tests/golden/reference_yaml/import_reference_yaml.py
These vectors pin nothing until a converted ANI-2x model file is supplied (ANI2X_MODEL=... pytest tests/test_reference_yaml.py);
the trained weights are not in the container."""
import glob
import json
import os

import yaml

REF = "/root/reference/tests/lammps-unittest"
HERE = os.path.dirname(os.path.abspath(__file__))


class Loader(yaml.SafeLoader):
    pass


Loader.add_constructor("!", lambda loader, node: loader.construct_scalar(node))


def table(text, skip_index):
    if not text or not text.strip():   # the cuaev suites carry no stress (the reference's cuaev path returns a zero virial)
        return [None]
    rows = [[float(t) for t in line.split()] for line in text.strip().splitlines() if line.strip()]
    return [r[1:] for r in rows] if skip_index else rows


out = []
for path in sorted(glob.glob(os.path.join(REF, "*", "*.yaml"))):
    d = yaml.load(open(path), Loader=Loader)
    ps = d["pair_style"].split()   # ani cutoff model device num_models aev nbr precision
    out.append({
        "suite": os.path.basename(os.path.dirname(path)), "name": os.path.basename(path)[:-5],
        "cutoff": float(ps[1]), "device": ps[3], "num_models": int(ps[4]), "aev": ps[5], "nbr": ps[6], "precision": ps[7],
        "periodic": "boundary f f f" not in d.get("pre_commands", ""),
        "newton_bond_on": "newton_bond index on" in d.get("pre_commands", ""),
        "epsilon": float(d["epsilon"]), "natoms": int(d["natoms"]), "timestep_fs": 0.1, "run_steps": 4,
        "init_vdwl": float(d["init_vdwl"]), "run_vdwl": float(d["run_vdwl"]),
        "init_stress": table(d["init_stress"], False)[0], "run_stress": table(d["run_stress"], False)[0],
        "init_forces": table(d["init_forces"], True), "run_forces": table(d["run_forces"], True),
    })
json.dump(out, open(os.path.join(HERE, "reference_yaml.json"), "w"), indent=0)
print(f"{len(out)} fixtures:")
for o in out:
    print(f"  {o['suite']}/{o['name']}: {o['aev']} {o['nbr']} {o['precision']} {'pbc' if o['periodic'] else 'open'} eps {o['epsilon']} E0 {o['init_vdwl']}")
