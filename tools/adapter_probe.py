"""Where the host-pointer step goes: ani_compute_full on persistent caller arrays, pageable against page-locked
(ani_host_register), against the device-resident call; then the adapter's own host loops (numpy stand-ins)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg; _pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
p = "/tmp/probe.anim"
mf.write_model(p, mf.synthetic_model("ani2x", 1, seed=2024))
inp = hx.decompose(hx.spatial_sort(hx.water_box(n)))
lib = ani_hip.lib()
ani = ani_hip.ANI(p, 0)
nt, nl = inp.ntotal, inp.nlocal
species = np.ascontiguousarray(inp.species, dtype=np.int64)
x = np.ascontiguousarray(inp.x, dtype=np.float64)
il = np.ascontiguousarray(inp.ilist, dtype=np.int32); nn = np.ascontiguousarray(inp.numneigh, dtype=np.int32)
jl = np.ascontiguousarray(inp.jlist, dtype=np.int32)
e = np.zeros(1); f = np.zeros((nt, 3)); vir = np.zeros(9)


def step(ago):
    rc = lib.ani_compute_full(ani._h, nt, nl, species.ctypes.data, x.ctypes.data, inp.npairs, il.ctypes.data, jl.ctypes.data,
                              nn.ctypes.data, ago, 0, 0, e.ctypes.data, f.ctypes.data, None, vir.ctypes.data)
    assert rc == 0, lib.ani_last_error(ani._h)


def timed(label):
    step(0)
    for k in range(5):
        step(1 + k)
    t0 = time.perf_counter()
    for k in range(30):
        step(1 + k)
    t1 = time.perf_counter()
    for k in range(3):
        step(0)
    t2 = time.perf_counter()
    print(f"{label}: plain step {1e3 * (t1 - t0) / 30:.3f} ms, re-neighbouring step {1e3 * (t2 - t1) / 3:.3f} ms")


timed("pageable caller arrays   ")
r1 = lib.ani_host_register(x.ctypes.data, x.nbytes); r2 = lib.ani_host_register(f.ctypes.data, f.nbytes)
timed(f"page-locked x and f ({r1},{r2})")
r3 = lib.ani_host_register(jl.ctypes.data, jl.nbytes)
timed(f"+ page-locked jlist ({r3})   ")
for a in (x, f, jl):
    lib.ani_host_unregister(a.ctypes.data)
# the adapter's host loops (single thread): f += out_force over ntotal, reverse pack/unpack over the ghosts
ff = np.zeros((nt, 3)); t0 = time.perf_counter()
for k in range(10):
    ff += f
t1 = time.perf_counter()
print(f"f += out_force (numpy, {nt} atoms): {1e3 * (t1 - t0) / 10:.3f} ms")
