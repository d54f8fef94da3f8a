"""Hot-path phase times of the eight-member configurations under MLP launch variants.
usage: python tools/members_probe.py "opt=val,opt=val" ["opt=val,..." ...]   (each argument = one variant; "" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _pkg; _pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

dev = torch.device("cuda:0")
variants = sys.argv[1:] or [""]
cases = [("water-10002 x 8", "ani2x", False, hx.spatial_sort(hx.water_box(10002, seed=12345)))]
if not os.environ.get("MEMBERS_PROBE_FIRST_ONLY"):
    cases.append(("CH4/O2-100008 x 8 + repulsion", "ani1x", True, hx.spatial_sort(hx.combustion_box(100008, seed=12345))))
for name, kind, rep, sysm in cases:
    path = f"/tmp/mp_{kind}.anim"
    mf.write_model(path, mf.synthetic_model(kind, 8, seed=2024, repulsion=rep))
    inp = hx.decompose(sysm)
    d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
    d_sp = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
    d_il = torch.from_numpy(inp.ilist).to(dev); d_nn = torch.from_numpy(inp.numneigh).to(dev); d_jl = torch.from_numpy(inp.jlist).to(dev)
    d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev); d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for var in variants:
        ani = ani_hip.ANI(path, 0)
        for kv in filter(None, var.split(",")):
            k, v = kv.split("=")
            ani.set_option(k, int(v))
        def step(ago):
            d_f.zero_()
            ani.compute_device(inp.ntotal, inp.nlocal, d_sp.data_ptr(), d_x.data_ptr(), inp.npairs, d_il.data_ptr(), d_jl.data_ptr(),
                               d_nn.data_ptr(), ago, d_f.data_ptr(), d_ev.data_ptr(), stream=st)
        step(0)
        for k in range(5):
            step(1 + k)
        torch.cuda.synchronize()
        ani.phase_timing(1)
        t0 = time.perf_counter()
        for k in range(20):
            step(6 + k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        ph = ani.phase_times()
        c = max(ph["calls"], 1)
        print(f"{name:32s} [{var or 'defaults':28s}] step {dt:.3f} ms  mlp {ph['mlp'] / c:.3f}  aev_fwd {ph['aev_fwd'] / c:.3f}  aev_bwd {ph['aev_bwd'] / c:.3f}  "
              f"kernel {ani.last_mlp_kernel()}  E {float(d_ev[0]):.4f}")
        ani.close()
