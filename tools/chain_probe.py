"""Development probe: MLP phase time with the chained launch forced on/off at a given size.
   python tools/chain_probe.py [atoms] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
import numpy as np, torch
import bench

atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100002
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
wl = bench.Workload(atoms, 1, "cuaev", 0, 1, dev, 0, 0)
for mode in (1, 2, 0, 2, 1):
    wl.ani.set_option("mlp_chain", mode)
    dt, ph = wl.timed_run(steps, 5)
    c = max(ph["calls"], 1)
    print(f"mlp_chain={mode}: {dt / steps * 1e3:.4f} ms/step  mlp {ph['mlp'] / c:.4f}  fwd {ph['aev_fwd'] / c:.4f}  bwd {ph['aev_bwd'] / c:.4f}", flush=True)
f = wl.d_f.clone()
wl.ani.set_option("mlp_chain", 0); wl.step(1); torch.cuda.synchronize(); f0 = wl.d_f.clone()
wl.ani.set_option("mlp_chain", 2); wl.step(1); torch.cuda.synchronize(); f2 = wl.d_f.clone()
print("max |f(chain) - f(per layer)| =", float((f2 - f0).abs().max()))
