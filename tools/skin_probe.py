"""Development probe: hot-path step time as a function of the neighbour-list skin (what a shorter inner list would buy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
import torch
import bench
from lammps_ani_amd import harness as hx

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
orig = hx.decompose
for skin in (2.0, 1.0, 0.5, 0.25):
    hx.decompose = lambda system, grid=(1, 1, 1), rank=0, cutoff=5.1, skin=2.0, _s=skin, **kw: orig(system, grid, rank, cutoff=cutoff, skin=_s, **kw)
    wl = bench.Workload(atoms, 1, "cuaev", 0, 1, torch.device("cuda", 0), 0, 0)
    dt, ph = wl.timed_run(40, 5)
    c = max(ph["calls"], 1)
    print(f"skin {skin}: npairs/atom {wl.inp.npairs / wl.inp.nlocal:.1f}  {dt / 40 * 1e3:.4f} ms/step  fwd {ph['aev_fwd'] / c:.4f}  mlp {ph['mlp'] / c:.4f}  bwd {ph['aev_bwd'] / c:.4f}", flush=True)
    wl.close()
