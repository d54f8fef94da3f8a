"""MLP phase time with whole items, searched half items and every item as halves (option mlp_fused_halves 0 / 1 / 2), per size (GPU).
usage: python tools/halves_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _pkg; _pkg.load()
from lammps_ani_amd import ani_hip, harness as hx, model_file as mf

dev = torch.device("cuda:0")
cases = [(12501, 1, "mlp_fused_rows=128"), (25002, 1, ""), (50001, 1, ""), (75000, 1, ""), (100002, 1, ""), (10002, 8, "")]
for natoms, members, extra in cases:
    path = f"/tmp/hp_{members}.anim"
    mf.write_model(path, mf.synthetic_model("ani2x", members, seed=2024))
    inp = hx.decompose(hx.spatial_sort(hx.water_box(natoms, seed=12345)))
    d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
    d_sp = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
    d_il = torch.from_numpy(inp.ilist).to(dev); d_nn = torch.from_numpy(inp.numneigh).to(dev); d_jl = torch.from_numpy(inp.jlist).to(dev)
    d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev); d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    for halves in (0, 1, 2):
        ani = ani_hip.ANI(path, 0)
        ani.set_option("mlp_fused_halves", halves)
        for kv in filter(None, extra.split(",")):
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
        f = d_f.clone()
        ani.phase_timing(1)
        for k in range(30):
            step(6 + k)
        torch.cuda.synchronize()
        ph = ani.phase_times()
        c = max(ph["calls"], 1)
        if ref is None:
            ref = f
        print(f"{natoms:7d} atoms x {members}  halves {halves}  mlp {ph['mlp'] / c:.4f} ms  kernel {ani.last_mlp_kernel():20s} "
              f"max |dF| to whole items {float((f - ref).abs().max()):.2e}  E {float(d_ev[0]):.3f}", flush=True)
        ani.close()
