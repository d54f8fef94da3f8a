"""Scratch probe: per-step cost of the host-pointer entry point (ani_compute_full: H2D positions, D2H forces) vs the
device-resident one, and the cost of a list rebuild (ago = 0), on the benchmark box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg; _pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
p = "/tmp/probe.anim"
mf.write_model(p, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.spatial_sort(hx.water_box(n)))
ani = ani_hip.ANI(p, 0)
for flags in ((True, True), (False, False)):
    ani.compute(inp, ago=0, eflag_atom=flags[0], vflag=flags[1])
    t0 = time.perf_counter()
    for k in range(10):
        ani.compute(inp, ago=1 + k, eflag_atom=flags[0], vflag=flags[1])
    t1 = time.perf_counter()
    for k in range(3):
        ani.compute(inp, ago=0, eflag_atom=flags[0], vflag=flags[1])
    t2 = time.perf_counter()
    print(f"eflag_atom/vflag={flags}: host API step {1e3 * (t1 - t0) / 10:.3f} ms, rebuild step {1e3 * (t2 - t1) / 3:.3f} ms "
          f"(ntotal {inp.ntotal}, npairs {inp.npairs})")

# re-neighbouring step of the `devlist` adapter mode: species + positions up, list built on the device, then the step
ani.compute(inp, ago=0, eflag_atom=False, vflag=False)
t0 = time.perf_counter()
for k in range(5):
    npairs = ani.build_list(inp.species, inp.x, inp.nlocal, 7.1)
    ani.compute(inp, ago=1, eflag_atom=False, vflag=False)
t1 = time.perf_counter()
print(f"devlist rebuild step (ani_build_list + step): {1e3 * (t1 - t0) / 5:.3f} ms, npairs {npairs}")
sp64 = np.ascontiguousarray(inp.species, dtype=np.int64); xx = np.ascontiguousarray(inp.x, dtype=np.float64)
lo = xx.min(0) - 0.25; hi = xx.max(0) + 0.25
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(5):
        ani.build_list(sp64, xx, inp.nlocal, 7.1, lo, hi)
    t1 = time.perf_counter()
    print(f"ani_build_list alone (bounds given): {1e3 * (t1 - t0) / 5:.3f} ms")
t0 = time.perf_counter(); xx.min(0); xx.max(0); t1 = time.perf_counter()
print(f"numpy min/max: {1e3 * (t1 - t0):.3f} ms")
