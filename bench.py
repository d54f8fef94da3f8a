#!/usr/bin/env python3
"""bench.py — throughput of the ANI pair-style hot path on MI355X, one process per GPU.

A "step" is one pass of the hot path (what PairANI::compute does each MD step, src/pair_ani.cpp:66-233) over a
synthetic water box resident in HBM: zero forces -> AEV forward -> MLP ensemble forward/backward (fp32 MFMA) ->
AEV backward (forces on local+ghost atoms) -> ghost-force reverse exchange (index_add on one rank, RCCL
all_to_all_single between ranks).  ns/day = steps/s * 0.0432 at the reference's 0.5 fs timestep
(examples/benchmark/run_one.py:100, read_perf.py:26-32).  The neighbour list is built once (ago = 0, untimed) and
reused (ago > 0), positions are static: integration and list rebuilds are LAMMPS core work outside this path.

N > 1 (launched by torch.distributed.run): the SAME box is split into N bricks (strong scaling), as LAMMPS'
spatial decomposition does for the reference (examples/benchmark/submit_scaling.py:13-21).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import _pkg  # noqa: E402

_pkg.load()
from lammps_ani_amd import ani_hip, comm, harness as hx, model_file as mf  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0          # same table (spec; ~6.3 TB/s achievable)
# published: 100 002-atom water, ANI-2x, 1 model, fp32, 1xA100 (examples/benchmark/README.md:78)
PUBLISHED_NS_DAY = {(100002, 1): 1.495}


def mlp_flops_per_step(model, counts, aev_cols=None):
    """4 * M * sum_s n_s * P_s (forward 2P + input-gradient backward 2P), SURVEY.md §8(d).
    aev_cols: first-layer width actually contracted (the AEV columns of the species present); None = full AEV."""
    tot = 0
    for s, n in enumerate(counts):
        d = list(model.dims[s])
        if aev_cols is not None:
            d[0] = aev_cols
        P = sum(d[l] * d[l + 1] for l in range(len(d) - 1))
        tot += n * P
    return 4.0 * model.num_models * tot


def aev_bytes_per_step(A, nlocal, ntotal, npairs):
    """Algorithmic HBM bytes of AEV forward + backward, SURVEY.md §8(d)."""
    fwd = 4 * A * nlocal + 4 * npairs + 16 * ntotal + 8 * nlocal
    bwd = 4 * A * nlocal + 4 * npairs + 16 * ntotal + 12 * ntotal
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--atoms", type=int, default=100002, help="water-box size (multiple of 3)")
    ap.add_argument("--models", type=int, default=1, help="ensemble members used (ANI-2x has 8)")
    ap.add_argument("--aev", default="cuaev", choices=["cuaev", "pyaev"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vflag", type=int, default=0)
    ap.add_argument("--dense-aev", action="store_true", help="keep the AEV columns of absent species (full 1008-wide rows)")
    ap.add_argument("--no-dense-pass", action="store_true", help="skip the extra timed pass with the full-width AEV")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback path)")
    dev_index = local_rank % torch.cuda.device_count()  # the reference maps local_rank % num_devices too (src/pair_ani.cpp:269-272)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("ANI_BENCH_BACKEND", "nccl")  # "gloo" = host-staged rehearsal of the multi-rank path on one GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- workload ---------------------------------------------------------------------------------------
    model = mf.synthetic_model("ani2x", args.models, seed=2024)
    mpath = f"/tmp/bench_ani2x_m{args.models}_r{rank}.anim"
    mf.write_model(mpath, model)
    system = hx.spatial_sort(hx.water_box(args.atoms, seed=12345))  # LAMMPS sorts atoms spatially (atom_modify sort)
    grid = comm.grid_for(world)
    inp = hx.decompose(system, grid, rank, cutoff=5.1, skin=2.0)
    ani = ani_hip.ANI(mpath, dev_index, -1, use_cuaev=(args.aev == "cuaev"), use_fullnbr=True, use_single=True)
    if args.dense_aev:
        ani.set_option("prune_absent_species", 0)

    d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
    d_species = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
    d_ilist = torch.from_numpy(inp.ilist).to(dev)
    d_numneigh = torch.from_numpy(inp.numneigh).to(dev)
    d_jlist = torch.from_numpy(inp.jlist).to(dev)
    d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev)
    d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
    ex = comm.GhostExchange(inp, system.boxhi - system.boxlo, dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(ago):
        d_f.zero_()
        ani.compute_device(inp.ntotal, inp.nlocal, d_species.data_ptr(), d_x.data_ptr(), inp.npairs, d_ilist.data_ptr(),
                           d_jlist.data_ptr(), d_numneigh.data_ptr(), ago, d_f.data_ptr(), d_ev.data_ptr(), None,
                           eflag_atom=False, vflag=bool(args.vflag), stream=stream)
        ex.reverse_add(d_f.view(-1, 3))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(nsteps):
        step(0)  # list upload + bucketing: rebuild work, untimed
        for w in range(args.warmup):
            step(w + 1)
        sync_all()
        ani.phase_timing(True)
        t0 = time.perf_counter()
        for k in range(nsteps):
            step(args.warmup + 1 + k)
        sync_all()
        dt_ = time.perf_counter() - t0
        ph = ani.phase_times()
        ani.phase_timing(False)
        return dt_, ph

    dense_pass = None
    if world == 1 and not args.dense_aev and not args.no_dense_pass:
        # secondary number: the same workload with the full 1008-wide AEV rows (columns of absent species kept)
        ani.set_option("prune_absent_species", 0)
        dtd, phd = timed_run(max(args.steps // 2, 1))
        dense_pass = {"ms_per_step": dtd / max(args.steps // 2, 1) * 1e3, "value": max(args.steps // 2, 1) / dtd * 0.0432,
                      "phase_ms": {k: phd[k] / max(phd["calls"], 1) for k in ("aev_fwd", "mlp", "aev_bwd")}}
        ani.set_option("prune_absent_species", 1)
    dt, phases = timed_run(args.steps)
    energy_local = float(d_ev[0].item())
    if not np.isfinite(energy_local):
        raise SystemExit("non-finite energy: LDS neighbour capacity exceeded or numerical failure")
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank phase times and work, gathered for the roofline of the slowest rank
        stats = torch.tensor([phases["aev_fwd"], phases["mlp"], phases["aev_bwd"], phases["other"], inp.nlocal, inp.ntotal,
                              inp.npairs], dtype=torch.float64, device=cdev)
        allstats = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allstats, stats)
        allstats = torch.stack(allstats).cpu().numpy()
    else:
        allstats = np.array([[phases["aev_fwd"], phases["mlp"], phases["aev_bwd"], phases["other"], inp.nlocal, inp.ntotal,
                              inp.npairs]])

    if rank == 0:
        steps = args.steps
        ms_per_step = dt / steps * 1e3
        ns_day = steps / dt * 0.0432
        # roofline on the slowest rank (max MLP time); counts of that rank are not gathered per species, so use the
        # whole-box species counts scaled by its share of local atoms (water: H:O = 2:1 everywhere)
        r = int(np.argmax(allstats[:, 1]))
        calls = max(phases["calls"], 1)
        t_fwd, t_mlp, t_bwd, t_other = (allstats[r, i] / calls for i in range(4))
        nlocal_r, ntotal_r, npairs_r = (int(allstats[r, i]) for i in (4, 5, 6))
        counts_box = np.bincount(system.types - 1, minlength=model.num_species)
        counts_r = counts_box * (nlocal_r / system.natoms)
        aev_cols = ani.debug_view().aev_active_length  # columns of the species present (1008 when all 7 occur)
        flops = mlp_flops_per_step(model, counts_r, aev_cols)
        flops_dense = mlp_flops_per_step(model, counts_r)
        bf, bb = aev_bytes_per_step(aev_cols, nlocal_r, ntotal_r, npairs_r)
        mlp_roof = dict(bound="mfma", achieved=flops / (t_mlp * 1e-3) / 1e12 if t_mlp > 0 else None, peak=PEAK_F32_MFMA_TFLOPS,
                        unit="TFLOP/s", traffic=None, kernel="gemm_kernel (MLP forward+backward, 6 launches per species)",
                        ms_per_step=t_mlp, flops_per_step=flops, flops_per_step_full_width_aev=flops_dense, aev_columns=aev_cols)
        mlp_roof["frac"] = mlp_roof["achieved"] / PEAK_F32_MFMA_TFLOPS if mlp_roof["achieved"] else None
        aev_roof = dict(bound="hbm", achieved=(bf + bb) / ((t_fwd + t_bwd) * 1e-3) / 1e9 if t_fwd + t_bwd > 0 else None,
                        peak=PEAK_HBM_GBS, unit="GB/s", traffic=None, kernel="aev_forward_kernel + aev_backward_kernel",
                        ms_per_step=t_fwd + t_bwd, ms_fwd=t_fwd, ms_bwd=t_bwd, bytes_per_step=bf + bb)
        aev_roof["frac"] = aev_roof["achieved"] / PEAK_HBM_GBS if aev_roof["achieved"] else None
        dominant, other = (mlp_roof, aev_roof) if t_mlp >= t_fwd + t_bwd else (aev_roof, mlp_roof)

        out = {
            "metric": "MD ns/day for ANI-2x water box (0.5 fs steps; hot-path steps/s * 0.0432)",
            "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": (ns_day / PUBLISHED_NS_DAY[(args.atoms, args.models)]) if (world == 1 and (args.atoms, args.models) in PUBLISHED_NS_DAY) else None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"water-{args.atoms} (rho=0.98 g/cm3), ANI-2x shaped seeded weights, {args.models} model(s), "
                                   f"pair_style ani 5.1 <model> hip {args.models} {args.aev} full single, skin 2.0, static positions, list reused (ago>0)",
                       "atoms": args.atoms, "models": args.models, "grid": list(grid), "nlocal_rank0": inp.nlocal,
                       "nghost_rank0": inp.nghost, "npairs_rank0": inp.npairs, "aev": args.aev, "vflag": args.vflag, "prune_absent_species": not args.dense_aev, "aev_columns": aev_cols,
                       "matom_steps_per_s": args.atoms * steps / dt / 1e6,
                       "vs_baseline_note": "published number is 1xA100 (examples/benchmark/README.md:78), different hardware"},
            "roofline": dominant, "roofline_other": other, "full_width_aev_pass": dense_pass,
            "phase_ms": {"aev_fwd": t_fwd, "mlp": t_mlp, "aev_bwd": t_bwd, "finish": t_other},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import Oracle
            o = Oracle(mpath)  # fp64 restatement, OpenMP over all host cores
            o.compute(inp) if inp.nlocal <= 20000 else None  # warm caches on small inputs only
            tc = time.perf_counter()
            ref = o.compute(inp, radial_compat=(args.aev == "pyaev"))
            tcpu = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": 0.0432 / tcpu, "unit": "ns/day", "cores": o.threads, "kind": "port",
                                   "sample": f"1 force evaluation of the same {args.atoms}-atom workload with oracle/ani_oracle.c (fp64, OpenMP), {tcpu:.2f} s",
                                   "ms_per_step": tcpu * 1e3}
            # parity of the benchmarked configuration itself (forces of the last step vs the oracle)
            f = d_f.view(-1, 3).cpu().numpy()
            fr = ref["force"][: inp.nlocal].copy()
            np.add.at(fr, inp.owner_lidx, ref["force"][inp.nlocal:])
            err = np.abs(f[: inp.nlocal] - fr)
            out["parity"] = {"max_abs_force_err_kcal_mol_A": float(err.max()),
                             "p999_abs_force_err": float(np.percentile(err, 99.9)),
                             "rms_force_err": float(np.sqrt((err ** 2).mean())),
                             "max_abs_force": float(np.abs(fr).max()), "rms_force": float(np.sqrt((fr ** 2).mean())),
                             "energy_err_kcal_mol": float(abs(energy_local - ref["energy"])),
                             "tolerance_note": "north_star bar: 1e-4 eV/A = 2.3e-3 kcal/mol/A"}
        print(json.dumps(out))
    ani.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
