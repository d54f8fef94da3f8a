"""Imports the known-answer vector embedded in the reference's C++ test (src/ani_csrc/test_model.cpp:84-137) into a
JSON fixture.  Data only — the numbers of the four brace-initialised arrays and the expected energy — nothing else of
the file:

  coords[30*3]        the 30-atom water cluster (same coordinates as tests/water-0.8nm.data), open boundary
  species[30]         O = 3, H = 0
  atom_index12[2*411] the HALF neighbour list the reference's LAMMPS produced for it at 5.1 + 2.0 = 7.1 A
                      ([0:n] = i, [n:2n] = j): 411 of the 435 pairs
  expected_force[90]  fp64 forces of the trained 8-member ANI-2x ensemble, kcal/mol/A
  expected_energy     Hartree

The list needs no trained weights to be checked: it pins the neighbour-list builders (host harness,
ani_build_list, ani_build_list_device) as pair sets (tests/test_reference_kat.py).  The energy / forces stay
unreachable until ANI-2x weights are supplied.

Run where /root/reference exists:  python tests/golden/reference_kat/import_reference_kat.py"""
import json
import os
import re

SRC = "/root/reference/src/ani_csrc/test_model.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))

text = open(SRC).read()
body = text[text.index("int test_ani2x_withnbr"):]


def array(name):
    m = re.search(r"std::vector<[^>]+>\s+" + name + r"\s*=\s*\{([^}]*)\}", body)
    return [float(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()]


coords = array("coords")
species = [int(v) for v in array("species")]
a12 = [int(v) for v in array("atom_index12")]
force = array("expected_force")
energy = float(re.search(r"expected_energy\s*=\s*(-?[0-9.]+)\s*\*\s*hartree2kcalmol", body).group(1))
assert len(coords) == 90 and len(species) == 30 and len(force) == 90 and len(a12) % 2 == 0
out = {"source": "src/ani_csrc/test_model.cpp:84-137", "cutoff": 5.1, "skin": 2.0, "boundary": "open",
       "coords": coords, "species": species, "atom_index12": a12, "expected_force_kcal_mol_A": force,
       "expected_energy_hartree": energy, "threshold_fp64": 1e-8, "threshold_fp32": 3e-4}
json.dump(out, open(os.path.join(HERE, "test_model_kat.json"), "w"))
print(f"{len(a12) // 2} pairs, {len(species)} atoms, E = {energy} Ha")
