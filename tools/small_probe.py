"""Development probe: the step of a small per-GPU system (the 8-GPU share of the benchmark box), for the kernel timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
import torch
import bench

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 12501
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.Workload(atoms, 1, "cuaev", 0, 1, torch.device("cuda", 0), 0, 0)
wl.ani.phase_timing(0)
wl.step(0)
for k in range(20):
    wl.step(k + 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    wl.step(k + 1)
torch.cuda.synchronize()
print(f"{atoms} atoms: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step (no phase events)")
