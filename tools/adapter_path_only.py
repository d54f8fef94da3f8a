"""bench.py's adapter_path alone (GPU): python tools/adapter_path_only.py [natoms]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402

_pkg.load()
import bench  # noqa: E402
from lammps_ani_amd import harness as hx, model_file as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
p = "/tmp/adapter_only.anim"
mf.write_model(p, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.spatial_sort(hx.water_box(n)))
for _ in range(2):
    print(json.dumps(bench.adapter_path(inp, p), indent=1), flush=True)
