#!/usr/bin/env python3
"""bench.py — throughput of the ANI pair-style hot path on MI355X, one process per GPU.

A "step" is one pass of the hot path (what PairANI::compute does each MD step, src/pair_ani.cpp:66-233) over a
synthetic water box resident in HBM: zero forces -> AEV forward -> MLP ensemble forward/backward (fp32 via split-bf16 MFMA) ->
AEV backward (forces on local+ghost atoms) -> ghost-force reverse exchange (index_add on one rank, RCCL
all_to_all_single between ranks).  ns/day = steps/s * 0.0432 at the reference's 0.5 fs timestep
(examples/benchmark/run_one.py:100, read_perf.py:26-32).  The neighbour list is built once (ago = 0, untimed) and
reused (ago > 0), positions are static: integration and list rebuilds are LAMMPS core work outside this path.

Workload at N = 1: the 100 002-atom water box with 1 ensemble member — the configuration the reference publishes
(examples/benchmark/README.md:78).  N > 1 (launched by torch.distributed.run): the SAME box split into N bricks
(strong scaling), as LAMMPS' spatial decomposition does for the reference (examples/benchmark/submit_scaling.py:13-21).

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


class Workload:
    """One water box on this rank: system, decomposition, device tensors, library handle, ghost exchange."""

    def __init__(self, atoms, models, aev, rank, world, dev, dev_index, vflag, repulsion=False, kind="ani2x", system=None):
        self.atoms, self.models, self.world, self.vflag = atoms, models, world, vflag
        self.model = mf.synthetic_model(kind, models, seed=2024, repulsion=repulsion)
        self.mpath = f"/tmp/bench_{kind}_m{models}_r{rank}.anim"
        mf.write_model(self.mpath, self.model)
        # LAMMPS sorts atoms spatially (atom_modify sort): neighbours are then close in memory
        self.system = hx.spatial_sort(hx.water_box(atoms, seed=12345) if system is None else system)
        self.grid = comm.grid_for(world)
        self.inp = inp = hx.decompose(self.system, self.grid, rank, cutoff=5.1, skin=2.0)
        self.ani = ani_hip.ANI(self.mpath, dev_index, -1, use_cuaev=(aev == "cuaev"), use_fullnbr=True, use_single=True)
        self.d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
        self.d_species = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
        self.d_ilist = torch.from_numpy(inp.ilist).to(dev)
        self.d_numneigh = torch.from_numpy(inp.numneigh).to(dev)
        self.d_jlist = torch.from_numpy(inp.jlist).to(dev)
        self.d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev)
        self.d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
        self.ex = comm.GhostExchange(inp, self.system.boxhi - self.system.boxlo, dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self, ago):
        inp = self.inp
        self.d_f.zero_()
        self.ani.compute_device(inp.ntotal, inp.nlocal, self.d_species.data_ptr(), self.d_x.data_ptr(), inp.npairs,
                                self.d_ilist.data_ptr(), self.d_jlist.data_ptr(), self.d_numneigh.data_ptr(), ago,
                                self.d_f.data_ptr(), self.d_ev.data_ptr(), None, eflag_atom=False, vflag=bool(self.vflag),
                                stream=self.stream)
        self.ex.reverse_add(self.d_f.view(-1, 3))

    def sync_all(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(self, nsteps, warmup):
        self.step(0)  # list upload + bucketing: rebuild work, untimed
        for w in range(warmup):
            self.step(w + 1)
        self.sync_all()
        # the per-phase HIP events (5 records per step, each a ~5 us bubble on the stream) are sampled on every 4th
        # step only: at small per-GPU sizes they would otherwise cost several percent of the step being measured
        self.ani.phase_timing(1)   # fresh accumulation ...
        self.ani.phase_timing(0)   # ... recording paused
        t0 = time.perf_counter()
        for k in range(nsteps):
            if k % 4 == 0:
                self.ani.phase_timing(2)
            self.step(warmup + 1 + k)
            if k % 4 == 0:
                self.ani.phase_timing(0)
        self.sync_all()
        dt = time.perf_counter() - t0
        ph = self.ani.phase_times()
        return dt, ph

    def close(self):
        self.ani.close()


def md_loop_pass(system, models, aev, dev, dev_index, steps, warmup):
    """The whole timestep loop of the reference's benchmark input (examples/benchmark/in.lammps:24-26,54-72: velocity
    create 300 K, fix langevin 300 300 100 + fix nve, dt 0.5 fs, neighbor 2.0 bin, every 10 check yes) with everything
    on the device: lammps_ani_amd.md.VerletRun (integration, displacement checks, device neighbour-list rebuilds with
    ghost regeneration, ghost exchange) around the same hot path.  The seeded weights have no minimum at the start
    structure, so the output layer is scaled to keep the surface within a few kT (same shapes, same arithmetic)."""
    from lammps_ani_amd import md
    path = f"/tmp/bench_ani2x_m{models}_md.anim"
    mf.write_model(path, mf.synthetic_model("ani2x", models, seed=2024, out_scale=0.02))
    inp = hx.decompose(system, (1, 1, 1), 0, cutoff=5.1, skin=2.0)
    ani = ani_hip.ANI(path, dev_index, -1, use_cuaev=(aev == "cuaev"), use_fullnbr=True, use_single=True)
    run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo)
    run.create_velocities(300.0)
    for _ in range(warmup):
        run.step()
    torch.cuda.synchronize()
    b0, t0 = run.nbuilds, time.perf_counter()
    for _ in range(steps):
        run.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ke = run.kinetic_energy()
    out = {"what": "full MD loop on the device (integrate + langevin + neighbour rebuilds + ghost exchange + hot path)",
           "steps": steps, "ms_per_step": dt / steps * 1e3, "value": steps / dt * 0.0432, "unit": "ns/day",
           "list_rebuilds": run.nbuilds - b0, "npairs": run.npairs,
           "temperature_K": 2.0 * ke / (3.0 * run.nlocal - 3.0) / md.BOLTZ}
    ani.close()
    return out


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
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configuration (10 002 atoms, 8 members)")
    ap.add_argument("--no-md", action="store_true", help="skip the full-MD-loop pass (device neighbour list + integrator)")
    ap.add_argument("--repulsion", action="store_true", help="model with the optional pairwise repulsion block (not the headline configuration)")
    ap.add_argument("--md-steps", type=int, default=200)
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON); the library's load banner (printed to stdout like the reference's,
    # src/ani_csrc/ani.cpp:88-92) and anything else written to fd 1 goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    wl = Workload(args.atoms, args.models, args.aev, rank, world, dev, dev_index, args.vflag, args.repulsion)
    ani, inp, model, system = wl.ani, wl.inp, wl.model, wl.system
    if args.dense_aev:
        ani.set_option("prune_absent_species", 0)

    dense_pass = None
    if world == 1 and not args.dense_aev and not args.no_dense_pass:
        # secondary number: the same workload with the full 1008-wide AEV rows (columns of absent species kept)
        n2 = max(args.steps // 2, 1)
        ani.set_option("prune_absent_species", 0)
        dtd, phd = wl.timed_run(n2, args.warmup)
        dense_pass = {"ms_per_step": dtd / n2 * 1e3, "value": n2 / dtd * 0.0432,
                      "phase_ms": {k: phd[k] / max(phd["calls"], 1) for k in ("aev_fwd", "mlp", "aev_bwd")}}
        fl = mlp_flops_per_step(model, np.bincount(system.types - 1, minlength=model.num_species))
        dense_pass["mlp_tflops"] = fl / (dense_pass["phase_ms"]["mlp"] * 1e-3) / 1e12
        dense_pass["mlp_frac_of_f32_mfma_peak"] = dense_pass["mlp_tflops"] / PEAK_F32_MFMA_TFLOPS
        ani.set_option("prune_absent_species", 1)
    dt, phases = wl.timed_run(args.steps, args.warmup)
    energy_local = float(wl.d_ev[0].item())
    if not np.isfinite(energy_local):
        raise SystemExit("non-finite energy: LDS neighbour capacity exceeded or numerical failure")
    stats_row = [phases["aev_fwd"], phases["mlp"], phases["aev_bwd"], phases["other"], inp.nlocal, inp.ntotal, inp.npairs]
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        stats = torch.tensor(stats_row, dtype=torch.float64, device=cdev)
        allstats = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allstats, stats)
        allstats = torch.stack(allstats).cpu().numpy()
    else:
        allstats = np.array([stats_row])

    out = None
    if rank == 0:
        steps = args.steps
        ms_per_step = dt / steps * 1e3
        ns_day = steps / dt * 0.0432
        # roofline of the slowest rank (largest MLP time); species counts of that rank = whole-box counts scaled by its
        # share of the atoms (water: H:O = 2:1 everywhere)
        r = int(np.argmax(allstats[:, 1]))
        calls = max(phases["calls"], 1)
        t_fwd, t_mlp, t_bwd, t_other = (allstats[r, i] / calls for i in range(4))
        nlocal_r, ntotal_r, npairs_r = (int(allstats[r, i]) for i in (4, 5, 6))
        counts_r = np.bincount(system.types - 1, minlength=model.num_species) * (nlocal_r / system.natoms)
        aev_cols = ani.debug_view().aev_active_length  # columns of the species present (1008 when all 7 occur)
        flops = mlp_flops_per_step(model, counts_r, aev_cols)
        flops_dense = mlp_flops_per_step(model, counts_r)
        bf, bb = aev_bytes_per_step(aev_cols, nlocal_r, ntotal_r, npairs_r)
        mlp_roof = dict(bound="mfma", achieved=flops / (t_mlp * 1e-3) / 1e12 if t_mlp > 0 else None, peak=PEAK_F32_MFMA_TFLOPS,
                        unit="TFLOP/s", traffic=None,
                        kernel="gemm_grouped_x3 (MLP forward + backward: 6 grouped launches per step, all species and members)",
                        ms_per_step=t_mlp, flops_per_step=flops, flops_per_step_full_width_aev=flops_dense, aev_columns=aev_cols,
                        note="achieved = algorithmic fp32 flops / time, peak = the fp32-input MFMA peak.  The kernel evaluates each "
                             "fp32 product as six v_mfma_f32_32x32x16_bf16 products of the exact hi/mid/lo bf16 splits of both "
                             "operands (fp32 accumulate; same force error against the fp64 oracle as the fp32-input MFMA path, "
                             "option mlp_split_bf16=0), so the MFMA pipe executes 6x these flops at the bf16 rate")
        mlp_roof["frac"] = mlp_roof["achieved"] / PEAK_F32_MFMA_TFLOPS if mlp_roof["achieved"] else None
        # share of the step's MLP time the MFMA pipes are busy: 6 bf16 instructions of 32 cycles per 32x32x16 block
        mlp_roof["mfma_pipe_busy_frac"] = (flops * 6 / (2 * 32 * 32 * 16) * 32 / (1024 * 2.4e9)) / (t_mlp * 1e-3) if t_mlp > 0 else None
        aev_roof = dict(bound="hbm", achieved=(bf + bb) / ((t_fwd + t_bwd) * 1e-3) / 1e9 if t_fwd + t_bwd > 0 else None,
                        peak=PEAK_HBM_GBS, unit="GB/s", traffic=None, kernel="aev_forward_fast + aev_backward_fast",
                        ms_per_step=t_fwd + t_bwd, ms_fwd=t_fwd, ms_bwd=t_bwd, bytes_per_step=bf + bb,
                        note="issue-bound, not HBM-bound: ~3300 VALU wave-instructions per centre, forward + backward (DESIGN.md 3.1); "
                             "counters in profiles/r01_e_pmc_summary.json")
        aev_roof["frac"] = aev_roof["achieved"] / PEAK_HBM_GBS if aev_roof["achieved"] else None
        # HBM traffic per launch from the PMC passes of the same command (tools/profile_round.sh; FETCH_SIZE and
        # WRITE_SIZE in separate runs, KB; FETCH_SIZE doubled: gfx950 counts half of a wide read, MI355X_MICROARCH.md)
        pmc_file = os.path.join(ROOT, "profiles", "r01_e_pmc_summary.json")
        if world == 1 and (args.atoms, args.models) == (100002, 1) and not args.dense_aev and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))

            def hbm_bytes(prefix):
                tot = 0.0
                for name, c in pmc.items():
                    if name.startswith(prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                        per_step = c["FETCH_SIZE"]["launches"] / max(pmc["ani::pack_kernel"]["FETCH_SIZE"]["launches"], 1)
                        tot += (2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024.0 * per_step
                return tot
            aev_roof["traffic"] = hbm_bytes("ani::aev_")
            aev_roof["traffic_note"] = "bytes per step, forward + backward launch; profiles/r01_e_pmc_summary.json"
            mlp_roof["traffic"] = hbm_bytes("ani::gemm_grouped")
            mlp_roof["traffic_note"] = "bytes per step over the 6 launches; profiles/r01_e_pmc_summary.json"
        dominant, other = (mlp_roof, aev_roof) if t_mlp >= t_fwd + t_bwd else (aev_roof, mlp_roof)

        out = {
            "metric": "MD ns/day for ANI-2x water box (0.5 fs steps; hot-path steps/s * 0.0432)",
            "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": (ns_day / PUBLISHED_NS_DAY[(args.atoms, args.models)]) if (world == 1 and (args.atoms, args.models) in PUBLISHED_NS_DAY) else None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 in/out and accumulation everywhere; MLP products via exact 3-way bf16 operand splits (6 MFMA terms)",
            "config": {"workload": f"water-{args.atoms} (rho=0.98 g/cm3), ANI-2x shaped seeded weights, {args.models} model(s), "
                                   f"pair_style ani 5.1 <model> hip {args.models} {args.aev} full single, skin 2.0, static positions, list reused (ago>0)",
                       "atoms": args.atoms, "models": args.models, "grid": list(wl.grid), "nlocal_rank0": inp.nlocal,
                       "nghost_rank0": inp.nghost, "npairs_rank0": inp.npairs, "aev": args.aev, "vflag": args.vflag,
                       "prune_absent_species": not args.dense_aev, "aev_columns": aev_cols,
                       "matom_steps_per_s": args.atoms * steps / dt / 1e6,
                       "vs_baseline_note": "published number is 1xA100 (examples/benchmark/README.md:78), different hardware"},
            "roofline": dominant, "roofline_other": other, "full_width_aev_pass": dense_pass,
            "phase_ms": {"aev_fwd": t_fwd, "mlp": t_mlp, "aev_bwd": t_bwd, "finish": t_other},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import Oracle
            o = Oracle(wl.mpath)  # fp64 restatement, OpenMP over all host cores
            tc = time.perf_counter()
            ref = o.compute(inp, radial_compat=(args.aev == "pyaev"))
            tcpu = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": 0.0432 / tcpu, "unit": "ns/day", "cores": o.threads, "kind": "port",
                                   "sample": f"1 force evaluation of the same {args.atoms}-atom workload with oracle/ani_oracle.c (fp64, OpenMP), {tcpu:.2f} s",
                                   "ms_per_step": tcpu * 1e3}
            # parity of the benchmarked configuration itself (forces of the last step vs the oracle)
            f = wl.d_f.view(-1, 3).cpu().numpy()
            fr = ref["force"][: inp.nlocal].copy()
            np.add.at(fr, inp.owner_lidx, ref["force"][inp.nlocal:])
            err = np.abs(f[: inp.nlocal] - fr)
            out["parity"] = {"max_abs_force_err_kcal_mol_A": float(err.max()),
                             "p999_abs_force_err": float(np.percentile(err, 99.9)),
                             "rms_force_err": float(np.sqrt((err ** 2).mean())),
                             "max_abs_force": float(np.abs(fr).max()), "rms_force": float(np.sqrt((fr ** 2).mean())),
                             "energy_err_kcal_mol": float(abs(energy_local - ref["energy"])),
                             "tolerance_note": "north_star bar: 1e-4 eV/A = 2.3e-3 kcal/mol/A"}
            # the same step with the MLP on fp32-input MFMA (v_mfma_f32_32x32x2_f32) instead of the split-bf16 products
            ani.set_option("mlp_split_bf16", 0)
            dt32, ph32 = wl.timed_run(max(args.steps // 5, 1), 2)
            f32 = wl.d_f.view(-1, 3).cpu().numpy()
            e32 = np.abs(f32[: inp.nlocal] - fr)
            out["parity"]["fp32_input_mfma_path"] = {
                "max_abs_force_err_kcal_mol_A": float(e32.max()), "rms_force_err": float(np.sqrt((e32 ** 2).mean())),
                "max_abs_force_diff_to_split_path": float(np.abs(f32[: inp.nlocal] - f[: inp.nlocal]).max()),
                "mlp_ms_per_step": ph32["mlp"] / max(ph32["calls"], 1)}
            ani.set_option("mlp_split_bf16", 1)
    wl.close()

    if rank == 0 and world == 1 and not args.no_extra and (args.atoms, args.models) == (100002, 1):
        # BASELINE.json configs[1]: full 8-member ensemble on a ~10k-atom water box (not the headline value)
        w2 = Workload(10002, 8, args.aev, rank, world, dev, dev_index, args.vflag)
        dt2, ph2 = w2.timed_run(args.steps, args.warmup)
        c2 = max(ph2["calls"], 1)
        out["extra_config"] = {"workload": "water-10002, ANI-2x shaped, 8 models (BASELINE.json configs[1])",
                               "ms_per_step": dt2 / args.steps * 1e3, "value": args.steps / dt2 * 0.0432, "unit": "ns/day",
                               "phase_ms": {k: ph2[k] / c2 for k in ("aev_fwd", "mlp", "aev_bwd")}}
        w2.close()
        # BASELINE.json configs[4] shape on one GPU: reactive CH4:O2 gas (3 of the 4 ANI-1x species present, ~26 list
        # entries per atom), ANI-1x shaped 8-member ensemble with the pairwise repulsion of the reactive models
        sysm = hx.combustion_box(100008, seed=12345)
        w3 = Workload(len(sysm.x), 8, args.aev, rank, world, dev, dev_index, args.vflag, repulsion=True, kind="ani1x", system=sysm)
        dt3, ph3 = w3.timed_run(args.steps, args.warmup)
        c3 = max(ph3["calls"], 1)
        out["mixed_species_config"] = {
            "workload": f"CH4:O2 1:2 gas, 0.25 g/cm3, {len(sysm.x)} atoms (H,C,O of the 4 ANI-1x species; BASELINE.json configs[4] "
                        "shape on one GPU), ANI-1x shaped, 8 models, pairwise repulsion on",
            "ms_per_step": dt3 / args.steps * 1e3, "value": args.steps / dt3 * 0.0432, "unit": "ns/day",
            "npairs_per_atom": w3.inp.npairs / max(w3.inp.nlocal, 1), "aev_columns": w3.ani.debug_view().aev_active_length,
            "phase_ms": {k: ph3[k] / c3 for k in ("aev_fwd", "mlp", "aev_bwd")}}
        w3.close()
    if rank == 0 and world == 1 and not args.no_md:
        out["md_loop"] = md_loop_pass(system, args.models, args.aev, dev, dev_index, args.md_steps, 20)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
