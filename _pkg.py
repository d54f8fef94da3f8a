"""Import helper: the package directory is named ``lammps-ani_amd`` (hyphen), which ``import`` cannot spell."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "lammps_ani_amd" in sys.modules:
        return sys.modules["lammps_ani_amd"]
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    pkg = importlib.import_module("lammps-ani_amd")
    sys.modules["lammps_ani_amd"] = pkg
    for sub in ("model_file", "harness", "ani_hip", "comm"):
        mod = importlib.import_module(f"lammps-ani_amd.{sub}")
        sys.modules[f"lammps_ani_amd.{sub}"] = mod
        setattr(pkg, sub, mod)
    return pkg
