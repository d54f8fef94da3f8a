#!/usr/bin/env python3
"""bench.py — MD throughput of the ANI pair-style hot path on MI355X, one process per GPU.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment this process starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` as a child BEFORE touching the GPU
(the reference's scaling runs do `mpirun -np {num_gpus}`, examples/benchmark/run_one.py:48), relays rank 0's JSON line
and exits with the child's code.  Launched by torch.distributed.run directly it reads RANK / LOCAL_RANK / WORLD_SIZE.

The headline `value` is the rate of the WHOLE timestep loop of the reference's benchmark input
(examples/benchmark/in.lammps:24-27,54-72: velocity create 300 K, fix langevin 300 300 100 + fix nve, dt 0.5 fs,
neighbor 2.0 bin, neigh_modify every 10 check yes) with everything on the device — lammps_ani_amd.md.VerletRun:
integration, displacement checks, re-neighbouring (atom migration, ghost shell, device neighbour list), ghost exchange
(RCCL between ranks) and the hot path (AEV forward, MLP forward/backward, AEV backward) — W warm-up steps, then exactly K
timed steps between barriers, MAX over ranks.  ns/day = steps/s * 0.0432 (examples/benchmark/run_one.py:100).
`hot_path` is the same workload with static positions and a reused list (the pair style's compute() alone).

Workload: the 100 002-atom water box with 1 ensemble member — the configuration the reference publishes
(examples/benchmark/README.md:78-81).  N > 1: the SAME box split into N bricks (strong scaling), as LAMMPS' spatial
decomposition does for the reference (examples/benchmark/submit_scaling.py:13-21).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0          # same table (spec; ~6.3 TB/s achievable)
# published: 100 002-atom water, ANI-2x, 1 model, fp32, Kokkos, on 1 / 2 / 4 / 8 A100 (examples/benchmark/README.md:78-81)
PUBLISHED_NS_DAY = {1: 1.495, 2: 2.774, 4: 4.846, 8: 7.663}
PEAK_16BIT_MFMA_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak, same table: the pipe the split products run on
PMC_SUMMARY = "r04_pmc_summary.json"
MD_OUT_SCALE = 0.02   # output-layer scale of the MD pass's model file (see md_pass)


def kernel_sources():
    """every source of libani_hip.so: a change in any of them invalidates a committed counter summary"""
    import glob
    d = os.path.join(ROOT, "lammps-ani_amd", "csrc")
    return sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp")) + glob.glob(os.path.join(d, "*.h")))


def apply_env_options(ani):
    """Development knob: ANI_BENCH_OPTIONS="name=value,..." is passed to ani_set_option on every handle the bench makes."""
    for kv in filter(None, os.environ.get("ANI_BENCH_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        ani.set_option(k.strip(), int(v))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--atoms", type=int, default=100002, help="water-box size (multiple of 3)")
    ap.add_argument("--models", type=int, default=1, help="ensemble members used (ANI-2x has 8)")
    ap.add_argument("--aev", default="cuaev", choices=["cuaev", "pyaev"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vflag", type=int, default=0)
    ap.add_argument("--dense-aev", action="store_true", help="keep the AEV columns of absent species (full 1008-wide rows)")
    ap.add_argument("--no-dense-pass", action="store_true", help="skip the extra timed pass with the full-width AEV")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (10 002 atoms x 8 members, combustion box)")
    ap.add_argument("--overlap", action="store_true", help="MD loop: ghost exchanges on a second stream beside the rows without "
                    "ghosts (ani_step_begin / _ghosts_ready / _finish); also ANI_MD_OVERLAP=1")
    ap.add_argument("--no-md", action="store_true", help="hot path only: `value` is then the static-position rate (development runs)")
    ap.add_argument("--repulsion", action="store_true", help="model with the optional pairwise repulsion block (not the headline configuration)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """Parent of an N-rank run: start the ranks as a child process tree (never exec: this process may not replace itself
    once anything has touched the GPU, and nothing here has), relay the JSON line, pass the exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without printing a JSON line\n")
        return 1
    return p.returncode


def source_digest():
    """sha256 over the kernel sources: ties a committed PMC summary to the code it was measured on."""
    h = hashlib.sha256()
    for path in kernel_sources():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


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
    """Algorithmic HBM bytes of AEV forward and backward, SURVEY.md §8(d) (the per-step compaction kernel does the
    candidate walk for both passes now; its bytes stay booked where the survey's formula books them)."""
    fwd = 4 * A * nlocal + 4 * npairs + 16 * ntotal + 8 * nlocal
    bwd = 4 * A * nlocal + 4 * npairs + 16 * ntotal + 12 * ntotal
    return fwd, bwd


# ---- compute bound of the AEV passes ("useful work per centre") ---------------------------------------
# Lane-operations (one FMA, multiply, add or compare of one lane) and transcendentals (exp / log / cos / sqrt / rcp) the
# ALGORITHM needs, counted from SURVEY.md 8(a) rows a5 / a6 with nA x nZ = 8 x 4 angular and 16 radial terms:
#   radial entry (neighbour inside Rcr), forward: distance 6 + cutoff 3 + 16 x (shift, square, scale, weight, add) ~ 45 ops;
#       sqrt, cos, 16 exp = 18 transcendentals.  Backward: the same terms again + 16 x (weight with dE/dAEV, derivative) + the
#       chain to the two atoms ~ 60 ops, 18 transcendentals.
#   triple (pair of neighbours inside Rca), forward: dot product, cos(theta'), sin, 4 x (angle shift 2, power 2) + 8 x radial
#       factor 3 + 32 products + 32 adds ~ 75 ops (FMA-fused); 1 sqrt + 4 x (log, exp) + 8 exp = 17 transcendentals.
#       Backward: the factors again (43) + two 8x4 contractions with dE/dAEV and their derivatives (~100) + the chain rule to
#       three atoms (~45) ~ 190 ops, 17 transcendentals.
# Priced at the vector unit's measured issue rate: one wave64 fp32 instruction per 2.6 cycles and SIMD at saturation, a
# transcendental 5x that (tools/issue_probe.hip, profiles/r03_issue_probe.log), 1024 SIMDs at the nominal 2.4 GHz.  The
# datasheet's 157.3 TFLOP/s counts packed FMAs on every lane; the unpacked FMA rate behind this bound is 121 TFLOP/s.
AEV_OPS = {"fwd": {"radial": (45, 18), "triple": (75, 17)}, "bwd": {"radial": (60, 18), "triple": (190, 17)}}
VALU_CYCLES_PER_WAVE_INSTR = 2.6
TRANSCENDENTAL_COST = 5.0


def neighbour_statistics(inp, rcr, rca):
    """(radial entries, angular neighbours, triples) summed over the owned atoms, from the positions and the list"""
    import numpy as np
    i = np.repeat(np.arange(inp.nlocal), inp.numneigh)
    nr = np.zeros(inp.nlocal, dtype=np.int64)
    na = np.zeros(inp.nlocal, dtype=np.int64)
    step = 4_000_000
    for a in range(0, i.shape[0], step):
        ii, jj = i[a:a + step], inp.jlist[a:a + step]
        d2 = ((inp.x[jj] - inp.x[ii]) ** 2).sum(1)
        nr += np.bincount(ii[d2 < rcr * rcr], minlength=inp.nlocal)
        na += np.bincount(ii[d2 < rca * rca], minlength=inp.nlocal)
    return int(nr.sum()), int(na.sum()), int((na * (na - 1) // 2).sum())


def aev_compute_bound_ms(which, n_radial, n_triples):
    """time the useful work of one AEV pass would take at the vector unit's issue rate (see AEV_OPS)"""
    (ro, rt), (to, tt) = AEV_OPS[which]["radial"], AEV_OPS[which]["triple"]
    lane_ops = n_radial * (ro + TRANSCENDENTAL_COST * rt) + n_triples * (to + TRANSCENDENTAL_COST * tt)
    wave_instr = lane_ops / 64.0
    return wave_instr * VALU_CYCLES_PER_WAVE_INSTR / (1024 * 2.4e9) * 1e3


def adapter_path(inp, model_path, nsteps=40, every=20):
    """The path LAMMPS itself calls, timed: tests/mock_lammps (a stand-in for LAMMPS' Atom / Neighbor / Comm objects) ->
    PairANI::compute (lammps-ani_amd/csrc/pair_ani.cpp) -> ani_compute_full with HOST pointers -> forces back in atom->f.
    Persistent arrays, eflag = vflag = 0, a re-neighbouring call every `every` steps; only compute() is timed
    (tests/mock_lammps/driver.cpp mock_md_loop).  Two list sources: the flattened LAMMPS list and the device-built list; the
    third variant also sums the ghost forces on the device (keyword rcclcomm: a one-rank communicator, device copies) instead of
    through the host's reverse communication."""
    import ctypes as C
    import numpy as np
    mock_dir = os.path.join(ROOT, "tests", "mock_lammps")
    so = os.path.join(mock_dir, "libpair_ani_mock.so")
    try:
        subprocess.run(["make", "-C", mock_dir, "-s"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        pass
    if not os.path.exists(so):
        return {"error": "tests/mock_lammps/libpair_ani_mock.so is missing"}
    lib = C.CDLL(so)
    lib.mock_create.restype = C.c_void_p
    lib.mock_create.argtypes = [C.c_char_p, C.c_int]
    lib.mock_error.restype = C.c_char_p
    lib.mock_error.argtypes = [C.c_void_p]
    lib.mock_pair_style.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int]
    lib.mock_compute.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    lib.mock_md_loop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.mock_destroy.argtypes = [C.c_void_p]
    nt = inp.ntotal
    x = np.ascontiguousarray(inp.x)
    ty = np.ascontiguousarray(inp.types, dtype=np.int32)
    nn = np.ascontiguousarray(inp.numneigh, dtype=np.int32)
    jl = np.ascontiguousarray(inp.jlist, dtype=np.int32)
    ow = np.ascontiguousarray(inp.owner_lidx, dtype=np.int32)
    out = {"what": "PairANI::compute through the mock LAMMPS objects, host-pointer entry points (what `pair_style ani` costs per step "
                   "before LAMMPS' own integrate / comm / neighbour work); eflag = vflag = 0",
           "steps": nsteps, "reneighbour_every": every}
    for source in ("hostlist", "devlist", "devlist rcclcomm"):
        h = lib.mock_create(b"real", 0)
        args = ["5.1", model_path, "hip", "-1", "cuaev", "full", "single"] + source.split()
        arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
        if lib.mock_pair_style(h, len(args), arr, 7) != 0:
            out[source] = {"error": lib.mock_error(h).decode()}
            continue
        f = np.zeros((nt, 3)); e = np.zeros(1); v = np.zeros(6)
        rc = lib.mock_compute(h, inp.nlocal, inp.nghost, x.ctypes.data, ty.ctypes.data, nn.ctypes.data, jl.ctypes.data, ow.ctypes.data,
                              0, 1, 0, f.ctypes.data, e.ctypes.data, v.ctypes.data, None)
        ms = np.zeros(3)
        if rc == 0:
            rc = lib.mock_md_loop(h, 12, 4, ms.ctypes.data)            # warm-up, re-neighbouring calls included (first-use costs)
        if rc == 0:
            rc = lib.mock_md_loop(h, nsteps, every, ms.ctypes.data)
        out[source] = {"plain_ms_per_step": float(ms[0]), "reneighbour_ms_per_step": float(ms[1]),
                       "ns_per_day_plain": 0.0432 / (float(ms[0]) * 1e-3) if ms[0] > 0 else None} if rc == 0 else \
            {"error": lib.mock_error(h).decode()}
        lib.mock_destroy(h)
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(launch_ranks(args))

    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkg
    _pkg.load()
    from lammps_ani_amd import ani_hip, comm, harness as hx, md, model_file as mf

    # stdout carries exactly ONE line (the JSON); the library's load banner (printed to stdout like the reference's,
    # src/ani_csrc/ani.cpp:88-92) and anything else written to fd 1 goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ANI_BENCH_BACKEND", "nccl")  # "gloo" = host-staged rehearsal of the multi-rank path on one GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback path)")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit(f"{world} ranks but only {ndev} HIP device(s) visible: RCCL refuses two ranks on one device "
                         "(set ANI_BENCH_BACKEND=gloo to rehearse the multi-rank path on fewer cards)")
    dev_index = local_rank % ndev  # the reference maps local_rank % num_devices too (src/pair_ani.cpp:269-272)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    n_ranks_seen = 1
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)   # the first collective: every rank is really there
        n_ranks_seen = int(one.item())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    class Workload:
        """One box on this rank for the hot-path pass: decomposition, device tensors, library handle, ghost exchange."""

        def __init__(self, atoms, models, repulsion=False, kind="ani2x", system=None):
            self.model = mf.synthetic_model(kind, models, seed=2024, repulsion=repulsion)
            self.mpath = f"/tmp/bench_{kind}_m{models}_r{rank}.anim"
            mf.write_model(self.mpath, self.model)
            # LAMMPS sorts atoms spatially (atom_modify sort): neighbours are then close in memory
            self.system = hx.spatial_sort(hx.water_box(atoms, seed=12345) if system is None else system)
            self.grid = comm.grid_for(world)
            self.inp = inp = hx.decompose(self.system, self.grid, rank, cutoff=5.1, skin=2.0)
            self.ani = ani_hip.ANI(self.mpath, dev_index, -1, use_cuaev=(args.aev == "cuaev"), use_fullnbr=True, use_single=True)
            apply_env_options(self.ani)
            self.d_x = torch.from_numpy(inp.x.reshape(-1)).to(dev)
            self.d_species = torch.from_numpy(inp.species.astype(np.int32)).to(dev)
            self.d_ilist = torch.from_numpy(inp.ilist).to(dev)
            self.d_numneigh = torch.from_numpy(inp.numneigh).to(dev)
            self.d_jlist = torch.from_numpy(inp.jlist).to(dev)
            self.d_f = torch.zeros(inp.ntotal * 3, dtype=torch.float64, device=dev)
            self.d_ev = torch.zeros(10, dtype=torch.float64, device=dev)
            self.ex = comm.GhostExchange(inp, self.system.boxhi - self.system.boxlo, dev)
            self.stream = torch.cuda.current_stream().cuda_stream
            # one rank: the ghost-force sum of the pair style as the loop's own kernel (include/ani_md.h), forces overwritten by
            # the library -- no tensor operation inside the timed steps
            self.native_reverse = world == 1
            if self.native_reverse:
                self.ani.set_option("device_overwrite_forces", 1)
                self.d_owner = torch.from_numpy(inp.owner_lidx.astype(np.int64)).to(dev)

        def step(self, ago):
            inp = self.inp
            if not self.native_reverse:
                self.d_f.zero_()
            self.ani.compute_device(inp.ntotal, inp.nlocal, self.d_species.data_ptr(), self.d_x.data_ptr(), inp.npairs,
                                    self.d_ilist.data_ptr(), self.d_jlist.data_ptr(), self.d_numneigh.data_ptr(), ago,
                                    self.d_f.data_ptr(), self.d_ev.data_ptr(), None, eflag_atom=False, vflag=bool(args.vflag),
                                    stream=self.stream)
            if self.native_reverse:
                rc = ani_hip.lib().ani_md_reverse_ghosts(self.d_f.data_ptr(), self.d_owner.data_ptr(), inp.nlocal, inp.nghost, self.stream)
                if rc:
                    raise RuntimeError(f"ani_md_reverse_ghosts: hipError {rc}")
            else:
                self.ex.reverse_add(self.d_f.view(-1, 3))

        def timed_run(self, nsteps, warmup):
            self.ani.phase_timing(1)   # the event pool is made now
            self.ani.phase_timing(0)
            self.step(0)  # list upload + bucketing: rebuild work, untimed
            for w in range(warmup):
                self.step(w + 1)
            sync_all()
            # the per-phase HIP events (6 records per step, each a ~5 us bubble on the stream) are sampled on every 4th
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
            sync_all()
            dt = max_over_ranks(time.perf_counter() - t0)
            return dt, self.ani.phase_times()

        def close(self):
            self.ani.close()

    def md_pass(system, steps, warmup, arith=None, with_parity=False, measure_rebuilds=True):
        """The reference benchmark's timestep loop on the device, W untimed + K timed steps.  arith: mlp_arith of the pass
        (None = the library default, the exact split).  After the timed region the loop keeps running, untimed, in blocks of
        `every` steps (the cadence of the displacement check) until it has seen re-neighbourings, so that the cost of a
        rebuild step and the interval between rebuilds are measured whatever K was."""
        path = f"/tmp/bench_ani2x_m{args.models}_md_r{rank}.anim"
        # The seeded weights have no minimum at the start structure, so the output layer is scaled to keep the surface
        # within a few kT (same shapes, same arithmetic, a liquid that stays a liquid at 300 K)
        md_model = mf.synthetic_model("ani2x", args.models, seed=2024, out_scale=MD_OUT_SCALE)
        mf.write_model(path, md_model)
        grid = comm.grid_for(world)
        inp = hx.decompose(system, grid, rank, cutoff=5.1, skin=2.0)
        ani = ani_hip.ANI(path, dev_index, -1, use_cuaev=(args.aev == "cuaev"), use_fullnbr=True, use_single=True)
        if arith is not None:
            ani.set_option("mlp_arith", arith)
        apply_env_options(ani)
        if args.dense_aev:
            ani.set_option("prune_absent_species", 0)
        # several ranks over RCCL: the exchanges run inside libani_hip.so (include/ani_comm.h: grouped ncclSend / ncclRecv on
        # the compute stream); ANI_BENCH_NATIVE_COMM=0 keeps them on torch.distributed's all_to_all_single for comparison
        native, native_note = None, None
        want_native = os.environ.get("ANI_BENCH_NATIVE_COMM", "1")   # "try": attempt it whatever the backend (rehearsals)
        if world > 1 and (backend == "nccl" or want_native == "try") and want_native not in ("", "0"):
            try:
                native = ani_hip.NativeComm.from_torch(dev_index)   # every rank gets one, or every rank raises
            except Exception as exc:   # the exchange then goes through torch.distributed; said so in the JSON line
                native_note = f"ani_comm unavailable ({exc}); "
            ok = torch.tensor([1.0 if native is not None else 0.0], device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # all ranks or none
            if float(ok) == 0.0 and native is not None:
                native.close()
                native, native_note = None, "ani_comm unavailable on another rank; "
        run = md.VerletRun(ani, inp, system.boxhi - system.boxlo, dev, dt=0.5, langevin=(300.0, 100.0), box_lo=system.boxlo, grid=grid,
                           overlap=True if args.overlap else None, native_comm=native)
        parity = None
        if with_parity and world == 1:
            # the MD pass runs on its own model file (output layer scaled): its forces at the start structure against the
            # oracle on that same file (owned atoms keep their order on one rank; ghost forces already folded by the loop)
            from oracle import Oracle
            ref = Oracle(path).compute(inp, radial_compat=(args.aev == "pyaev"))
            fr = ref["force"][: inp.nlocal].copy()
            np.add.at(fr, inp.owner_lidx, ref["force"][inp.nlocal:])
            run._forces()                 # the pair style's forces alone (set-up had already added the thermostat's share)
            f0 = run.f[: run.nlocal].cpu().numpy()
            run._post_force()             # ... which is put back before the first step integrates
            err = np.abs(f0 - fr)
            parity = {"max_abs_force_err_kcal_mol_A": float(err.max()), "rms_force_err": float(np.sqrt((err ** 2).mean())),
                      "max_abs_force": float(np.abs(fr).max()),
                      "energy_err_kcal_mol": float(abs(run.potential_energy() - ref["energy"])),
                      "what": f"forces of the MD pass's model (out_scale={MD_OUT_SCALE}) at the start structure vs oracle/ani_oracle.c on the same file"}
        run.create_velocities(300.0)
        # first-use costs belong to set-up, not to whichever step meets them first: the displacement check (a host round trip
        # every 10th step only) and the phase events are exercised here; the W warm-up steps follow
        run.warm_paths()
        ani.phase_timing(1)
        ani.phase_timing(0)
        for _ in range(warmup):
            run.step()
        sync_all()
        ani.phase_timing(1)
        ani.phase_timing(0)
        b0, t0 = run.nbuilds, time.perf_counter()
        # `run K` (no per-step output): the phase events of the library are recorded on every fourth step only
        run.run(steps, before_forces=lambda k: ani.phase_timing(2) if k % 4 == 0 else None,
                after_forces=lambda k: ani.phase_timing(0) if k % 4 == 0 else None)
        sync_all()
        dt = max_over_ranks(time.perf_counter() - t0)
        ph = ani.phase_times()
        ke = run.kinetic_energy()
        natoms_all = run._allreduce_sum(torch.tensor([float(run.nlocal)], dtype=torch.float64, device=dev))
        info = {"steps": steps, "ms_per_step": dt / steps * 1e3, "list_rebuilds": run.nbuilds - b0,
                "npairs_rank0": run.npairs, "nlocal_rank0": run.nlocal, "nghost_rank0": run.ntotal - run.nlocal,
                "temperature_K": 2.0 * ke / (3.0 * natoms_all - 3.0) / md.BOLTZ, "exchange_overlap": bool(run._overlap),
                "exchange": None if world == 1 else ("ani_comm: grouped ncclSend/ncclRecv inside libani_hip.so" if native is not None else
                                                     (native_note or "") + f"torch.distributed all_to_all_single ({backend})"),
                "energy_finite": bool(np.isfinite(run.potential_energy())), "model_out_scale": MD_OUT_SCALE,
                "mlp_arith": arith if arith is not None else 1}
        if parity:
            info["parity_md_model"] = parity
        if measure_rebuilds:
            # untimed continuation, without a break in the loop: on until it has re-neighboured three times (at least 100
            # steps, at most 600).  Its rate is the loop's rate whatever K was; with the K-step window it also gives the cost
            # of a plain step and the surcharge of a re-neighbouring step (two windows, two unknowns).
            nb0, n_post, t1 = run.nbuilds, 0, time.perf_counter()
            while n_post < 600 and (run.nbuilds - nb0 < 3 or n_post < 100):
                run.run(run.every)
                n_post += run.every
            sync_all()
            t_post = max_over_ranks(time.perf_counter() - t1)
            r1, r2 = info["list_rebuilds"], run.nbuilds - nb0
            info["long_window"] = {"steps": n_post, "list_rebuilds": r2, "ms_per_step": t_post / n_post * 1e3}
            if r2 > 0:
                info["amortised_ms_per_step"] = t_post / n_post * 1e3
                info["rebuild_interval_steps"] = (steps + n_post) / (r1 + r2)
            # cost of a re-neighbouring step, measured directly: single steps bracketed by synchronisations, alternately a
            # plain one and one forced to re-neighbour (both carry the same pipeline-refill bubble; their difference is the
            # surcharge of exchange + borders + device list + re-bucketing)
            t_plain, t_reb = [], []
            for _ in range(5):
                for forced in (False, True):
                    for _w in range(3):
                        run.step()
                    sync_all()
                    ts = time.perf_counter()
                    run.step(force_rebuild=forced)
                    sync_all()
                    (t_reb if forced else t_plain).append(max_over_ranks(time.perf_counter() - ts))
            info["plain_ms_per_step"] = float(np.median(t_plain)) * 1e3
            info["rebuild_ms"] = float(np.median(t_reb)) * 1e3
            info["rebuild_surcharge_ms"] = info["rebuild_ms"] - info["plain_ms_per_step"]
            info["rebuild_note"] = ("single synchronised steps, median of 5: a plain step and a step forced to re-neighbour; "
                                    "amortised_ms_per_step is the rate of the long window, re-neighbourings included")
            # a sustained block, untimed for `value` but reported: long enough (about two seconds of back-to-back steps) that a
            # coarse outside sampler of GPU activity sees the run, and a second, independent reading of the loop's rate
            n_sus = int(min(max(2.0 / max(dt / steps, 1e-5), 500), 20000)) // run.every * run.every
            sync_all()
            nb1, t2 = run.nbuilds, time.perf_counter()
            run.run(n_sus)
            sync_all()
            t_sus = max_over_ranks(time.perf_counter() - t2)
            info["sustained_window"] = {"steps": n_sus, "list_rebuilds": run.nbuilds - nb1, "ms_per_step": t_sus / n_sus * 1e3,
                                        "ns_per_day": n_sus / t_sus * 0.0432, "wall_s": t_sus}
            info["energy_finite"] = info["energy_finite"] and bool(np.isfinite(run.potential_energy()))
        view = ani.debug_view()
        ani.close()
        if native is not None:
            native.close()
        return dt, ph, info, view.aev_active_length

    # ---- hot path on static positions (phase times for the secondary numbers, parity against the oracle) -------------
    wl = Workload(args.atoms, args.models, repulsion=args.repulsion)
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
                      "phase_ms": {k: phd[k] / max(phd["calls"], 1) for k in ("compact", "aev_fwd", "mlp", "aev_bwd")}}
        fl = mlp_flops_per_step(model, np.bincount(system.types - 1, minlength=model.num_species))
        dense_pass["mlp_algorithmic_fp32_tflops"] = fl / (dense_pass["phase_ms"]["mlp"] * 1e-3) / 1e12
        dense_pass["mlp_frac_of_16bit_mfma_peak"] = 6.0 * dense_pass["mlp_algorithmic_fp32_tflops"] / PEAK_16BIT_MFMA_TFLOPS
        ani.set_option("prune_absent_species", 1)
    dt_hot, ph_hot = wl.timed_run(args.steps, args.warmup)
    energy_local = float(wl.d_ev[0].item())
    if not np.isfinite(energy_local) and not os.environ.get("ANI_BENCH_ALLOW_NAN"):   # the variable: timing-only ablation builds
        raise SystemExit("non-finite energy: LDS neighbour capacity exceeded or numerical failure")
    aev_cols = ani.debug_view().aev_active_length  # columns of the species present (1008 when all 7 occur)
    mlp_kernel_name = ani.last_mlp_kernel()

    # ---- the MD loop: the headline -----------------------------------------------------------------------------------
    if args.no_md:
        dt, phases, md_info = dt_hot, ph_hot, None
    else:
        dt, phases, md_info, aev_cols = md_pass(system, args.steps, args.warmup, with_parity=not args.no_cpu_baseline)
        if not md_info["energy_finite"]:
            raise SystemExit("non-finite energy in the MD loop")
    nl, nt, npairs = (md_info["nlocal_rank0"], md_info["nlocal_rank0"] + md_info["nghost_rank0"], md_info["npairs_rank0"]) \
        if md_info else (inp.nlocal, inp.ntotal, inp.npairs)

    out = None
    if rank == 0:
        steps = args.steps
        ms_per_step = dt / steps * 1e3
        ns_day = steps / dt * 0.0432
        value_basis = "K timed steps of the MD loop"
        if md_info:
            value_basis += f" ({md_info['list_rebuilds']} re-neighbouring(s) among them)"
            if md_info["list_rebuilds"] == 0 and "amortised_ms_per_step" in md_info:
                # a window without a re-neighbouring would overstate the loop's rate: the headline then charges the measured
                # rebuild surcharge at the measured interval (timed_region_* keep what the K steps themselves took)
                md_info["timed_region_ms_per_step"], md_info["timed_region_value"] = ms_per_step, ns_day
                ms_per_step = md_info["amortised_ms_per_step"]
                ns_day = 0.0432 / (ms_per_step * 1e-3)
                value_basis = ("the K timed steps contained no re-neighbouring and would overstate the loop: value is the rate of the longer "
                               "window that follows them in the same run, re-neighbourings included (md_loop.long_window)")
        if phases["calls"] == 0:   # the cut step of the overlapped exchange records no phase events: the hot-path pass has them
            phases = ph_hot
        calls = max(phases["calls"], 1)
        t_cmp, t_fwd, t_mlp, t_bwd, t_other = (phases[k] / calls for k in ("compact", "aev_fwd", "mlp", "aev_bwd", "other"))
        counts = np.bincount(system.types - 1, minlength=model.num_species) * (nl / system.natoms)
        flops = mlp_flops_per_step(model, counts, aev_cols)
        flops_dense = mlp_flops_per_step(model, counts)
        bf, bb = aev_bytes_per_step(aev_cols, nl, nt, npairs)

        # counters of a committed rocprofv3 --pmc run count only if they were taken on exactly these kernel sources
        pmc, pmc_note = None, f"no PMC summary for these kernel sources (profiles/{PMC_SUMMARY} is absent or was measured on other code)"
        pmc_file = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if world == 1 and (args.atoms, args.models) == (100002, 1) and not args.dense_aev and os.path.exists(pmc_file):
            cand = json.load(open(pmc_file))
            if cand.get("source_digest") == source_digest():
                pmc, pmc_note = cand["kernels"], f"profiles/{PMC_SUMMARY} (same kernel sources: digest " + cand["source_digest"] + ")"

        def counter(prefix, name):
            if not pmc:
                return None
            vals = [c[name]["mean_per_launch"] * c[name].get("launches_per_step", 1) for k, c in pmc.items() if k.startswith(prefix) and name in c]
            return sum(vals) if vals else None

        def hbm_traffic(prefix):
            f, w = counter(prefix, "FETCH_SIZE"), counter(prefix, "WRITE_SIZE")
            # KB; FETCH_SIZE doubled: gfx950 counts half of a wide read (MI355X_MICROARCH.md, HBM)
            return (2.0 * f + w) * 1024.0 if f is not None and w is not None else None

        def valu(prefix, t_ms):
            n = counter(prefix, "SQ_INSTS_VALU")
            # share of the chip's plain-FMA issue slots (one wave64 v_fma_f32 per 2.4 cycles and SIMD, tools/pk_probe.hip);
            # transcendentals, DPP and integer instructions hold a slot longer, so the pipes are busier than this says
            return n * 2.4 / (1024 * 2.4e9 * t_ms * 1e-3) if n and t_ms > 0 else None

        def valu4(prefix, t_ms):
            # the same count at the four cycles a wave64 fp32 instruction is seen to hold a SIMD's vector unit in these
            # kernels (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.2 cycles; phase stamps + ISA counts), at the
            # nominal 2.4 GHz -- the chip clocks lower under load, so the units are busier still
            n = counter(prefix, "SQ_INSTS_VALU")
            return n * 4.0 / (1024 * 2.4e9 * t_ms * 1e-3) if n and t_ms > 0 else None

        # useful work of the AEV passes (this rank's share), for their compute bound
        n_rad, n_ang, n_tri = neighbour_statistics(inp, *ani.cutoffs())
        share = nl / max(inp.nlocal, 1)
        n_rad, n_tri = n_rad * share, n_tri * share
        aev_note = ("bound by the CUs' vector issue, not by HBM (DESIGN.md 3.1): `frac` is the HBM fraction SURVEY 8(d) asks for "
                    "(algorithmic bytes / time / 8 TB/s); compute_bound_ms = the useful lane-operations and transcendentals of "
                    "the pass (bench.py AEV_OPS: counts per radial entry and per triple x this structure's entries) at the "
                    "vector unit's measured issue rate (2.6 cycles per wave64 instruction and SIMD, a transcendental 5x, "
                    "1024 SIMDs, 2.4 GHz); compute_frac = compute_bound_ms / time; valu_frac_4cyc = SQ_INSTS_VALU x 4 cycles / "
                    "(1024 SIMDs x time x 2.4 GHz): how busy the vector units are with the instructions the kernel executes")
        cb_b, cb_f = aev_compute_bound_ms("bwd", n_rad, n_tri), aev_compute_bound_ms("fwd", n_rad, n_tri)
        bwd_roof = dict(bound="hbm", achieved=bb / (t_bwd * 1e-3) / 1e9 if t_bwd > 0 else None, peak=PEAK_HBM_GBS, unit="GB/s",
                        traffic=hbm_traffic("ani::aev_backward"), kernel="aev_backward_fast", ms_per_launch=t_bwd, bytes_per_launch=bb,
                        compute_bound_ms=cb_b, compute_frac=cb_b / t_bwd if t_bwd > 0 else None,
                        valu_frac=valu("ani::aev_backward", t_bwd), valu_frac_4cyc=valu4("ani::aev_backward", t_bwd),
                        radial_entries=n_rad, triples=n_tri, note=aev_note, counters=pmc_note)
        bwd_roof["frac"] = bwd_roof["achieved"] / PEAK_HBM_GBS if bwd_roof["achieved"] else None
        t_f = t_fwd + t_cmp
        fwd_roof = dict(bound="hbm", achieved=bf / (t_f * 1e-3) / 1e9 if t_f > 0 else None, peak=PEAK_HBM_GBS, unit="GB/s",
                        traffic=(hbm_traffic("ani::aev_forward") or 0) + (hbm_traffic("ani::nbr_compact") or 0) if pmc else None,
                        kernel=("aev_forward_fused" if t_cmp < 0.25 * t_fwd and t_cmp < 0.02 else "nbr_compact_kernel + aev_forward_fast"),
                        kernel_note="neighbour compaction + AEV forward in one launch" if t_cmp < 0.25 * t_fwd and t_cmp < 0.02 else None,
                        ms_per_launch=t_f, ms_compact=t_cmp, ms_forward=t_fwd, bytes_per_launch=bf,
                        compute_bound_ms=cb_f, compute_frac=cb_f / t_f if t_f > 0 else None,
                        valu_frac=valu("ani::aev_forward", t_fwd), valu_frac_4cyc=valu4("ani::aev_forward", t_fwd),
                        radial_entries=n_rad, triples=n_tri, note=aev_note, counters=pmc_note)
        fwd_roof["frac"] = fwd_roof["achieved"] / PEAK_HBM_GBS if fwd_roof["achieved"] else None
        # The MLP against its bound: the matrix pipe.  The library default evaluates an fp32 product EXACTLY as six bf16 MFMA
        # products (mlp_arith 1), so the pipe it runs on is the 16-bit one and executes 6x the algorithmic flops:
        # frac = products x algorithmic flops / time / 2.5 PFLOP/s.
        nprod = {0: 1, 1: 6, 2: 3}[md_info["mlp_arith"] if md_info else 1]
        mlp_tr = (hbm_traffic("ani::mlp_fused") or hbm_traffic("ani::mlp_pipeline") or hbm_traffic("ani::gemm_grouped") or
                  hbm_traffic("ani::mlp_chain"))
        mlp_peak = PEAK_16BIT_MFMA_TFLOPS if nprod > 1 else PEAK_F32_MFMA_TFLOPS
        mlp_exec_tf = flops * nprod / (t_mlp * 1e-3) / 1e12 if t_mlp > 0 else None
        mlp_roof = dict(bound="mfma", achieved=mlp_exec_tf, peak=mlp_peak, unit="TFLOP/s", traffic=mlp_tr,
                        kernel=mlp_kernel_name, ms_per_launch=t_mlp, flops_per_launch=flops * nprod,
                        algorithmic_fp32_flops=flops, algorithmic_fp32_flops_full_width_aev=flops_dense, aev_columns=aev_cols,
                        mfma_products_per_fp32_product=nprod,
                        algorithmic_fp32_tflops=flops / (t_mlp * 1e-3) / 1e12 if t_mlp > 0 else None,
                        hbm_frac=(mlp_tr / (t_mlp * 1e-3) / 1e9 / PEAK_HBM_GBS) if (mlp_tr and t_mlp > 0) else None,
                        note="MLP forward + backward of every species bucket and member in one launch; achieved = the MFMA flops "
                             "executed (algorithmic fp32 flops 4 M sum_s n_s P_s over the AEV columns in use x MFMA products per fp32 "
                             "product) / the launch's average duration; peak = the dense peak of the pipe they run on "
                             "(/opt/skills/guides/MI355X_MICROARCH.md); hbm_frac = counter traffic / time / 8 TB/s",
                        counters=pmc_note)
        mlp_roof["frac"] = mlp_exec_tf / mlp_peak if mlp_exec_tf else None
        # counter view of the same thing: cycles the matrix pipes were busy / (4 SIMDs x cycles a CU was busy), while the launch ran
        mb, cb = counter("ani::mlp_fused", "SQ_VALU_MFMA_BUSY_CYCLES"), counter("ani::mlp_fused", "SQ_BUSY_CU_CYCLES")
        mlp_roof["mfma_busy_share"] = mb / (4.0 * cb) if mb and cb else None
        # `roofline` is the kernel with the largest share of the step (row 1 of the rocprofv3 --stats summary of this command)
        roofs = sorted([(t_mlp, mlp_roof), (t_bwd, bwd_roof), (t_f, fwd_roof)], key=lambda r: -r[0])
        main_roof, other_roofs = roofs[0][1], [r[1] for r in roofs[1:]]

        what = "static positions, list reused (hot path only)" if args.no_md else \
            ("velocity Verlet + Langevin 300 K, dt 0.5 fs, skin 2.0, rebuild check every 10 steps, all on the device: the "
             "device-resident stand-in loop md.VerletRun (the surface the reference reaches with pair_style ani/kk; the host-pointer "
             f"pair_style ani entry points add two PCIe copies per step, DESIGN.md section 6); MD pass on the model file with the "
             f"output layer scaled by {MD_OUT_SCALE} (same shapes and arithmetic, a liquid that stays a liquid)")
        out = {
            "metric": "MD ns/day for ANI-2x water box (0.5 fs steps; whole timestep loop: integrate + re-neighbour + ghost exchange + pair_style ani hot path)"
                      if not args.no_md else "hot-path-only ns/day for ANI-2x water box (NOT the MD metric: --no-md)",
            "value": ns_day, "unit": "ns/day", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "value_basis": value_basis, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": (ns_day / PUBLISHED_NS_DAY[world]) if (not args.no_md and (args.atoms, args.models) == (100002, 1) and world in PUBLISHED_NS_DAY) else None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 in/out and accumulation everywhere; the MLP's fp32 products are EXACT: both operands split into three bf16 "
                          "terms (8+8+8 mantissa bits), six bf16 MFMA products accumulated in fp32, the three dropped terms below "
                          "2^-23 of the product (library default mlp_arith = 1; the reference runs fp32 with TF32 off, "
                          "src/ani_csrc/ani.cpp:41-43).  `value_f16x2_split` is the same loop with the reduced-precision opt-in "
                          "(mlp_arith = 2, the counterpart of LAMMPS_ANI_ALLOW_TF32).  fp64 positions, velocities, energy sums",
            "config": {"workload": f"water-{args.atoms} (rho=0.98 g/cm3), ANI-2x shaped seeded weights, {args.models} model(s), "
                                   f"pair_style ani 5.1 <model> hip {args.models} {args.aev} full single; {what}",
                       "atoms": args.atoms, "models": args.models, "grid": list(comm.grid_for(world)), "nlocal_rank0": nl,
                       "nghost_rank0": nt - nl, "npairs_rank0": npairs, "aev": args.aev, "vflag": args.vflag,
                       "prune_absent_species": not args.dense_aev, "aev_columns": aev_cols, "backend": backend if world > 1 else None,
                       "matom_steps_per_s": args.atoms * steps / dt / 1e6,
                       "vs_baseline_note": "published numbers are LAMMPS Performance lines on A100s (examples/benchmark/README.md:78-81): other hardware, same metric and workload"},
            "roofline": main_roof, "roofline_other": other_roofs, "full_width_aev_pass": dense_pass,
            "phase_ms": {"nbr_compact": t_cmp, "aev_fwd": t_fwd, "mlp": t_mlp, "aev_bwd": t_bwd, "finish": t_other},
            "md_loop": md_info,
            "hot_path": {"what": "pair_style compute() alone: static positions, list reused (ago > 0), ghost-force exchange included",
                         "ms_per_step": dt_hot / steps * 1e3, "value": steps / dt_hot * 0.0432, "unit": "ns/day",
                         "phase_ms": {k: ph_hot[k] / max(ph_hot["calls"], 1) for k in ("compact", "aev_fwd", "mlp", "aev_bwd", "other")}},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import Oracle
            o = Oracle(wl.mpath)  # fp64 restatement, OpenMP over all host cores
            ref = o.compute(inp, radial_compat=(args.aev == "pyaev"))   # warm-up (page faults, thread pool), kept for the parity check
            ts = []
            for _ in range(3):
                tc = time.perf_counter()
                o.compute(inp, radial_compat=(args.aev == "pyaev"))
                ts.append(time.perf_counter() - tc)
            tcpu = float(np.median(ts))
            out["cpu_baseline"] = {"value": 0.0432 / tcpu, "unit": "ns/day", "cores": o.threads, "kind": "port",
                                   "sample": f"force evaluations of the same {args.atoms}-atom workload with oracle/ani_oracle.c (fp64, OpenMP): "
                                             f"1 warm-up, then median of 3 ({', '.join(f'{t:.2f}' for t in ts)} s); the hot path only, no integration",
                                   "ms_per_step": tcpu * 1e3}
            # parity of the benchmarked configuration itself (forces of the last hot-path step vs the oracle)
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
            # the same step with the other two arithmetics of the MLP: the exact three-term bf16 split (six products) and the
            # fp32-input MFMA instruction (v_mfma_f32_32x32x2_f32)
            out["parity"]["mlp_arith"] = "1: six bf16 MFMA products of the exact three-term splits (library default)"
            for key, arith in (("f16x2_split_path", 2), ("fp32_input_mfma_path", 0)):
                ani.set_option("mlp_arith", arith)
                dt32, ph32 = wl.timed_run(max(args.steps // 5, 1), 2)
                f32 = wl.d_f.view(-1, 3).cpu().numpy()
                e32 = np.abs(f32[: inp.nlocal] - fr)
                out["parity"][key] = {
                    "max_abs_force_err_kcal_mol_A": float(e32.max()), "p999_abs_force_err": float(np.percentile(e32, 99.9)),
                    "rms_force_err": float(np.sqrt((e32 ** 2).mean())),
                    "max_abs_force_diff_to_default_path": float(np.abs(f32[: inp.nlocal] - f[: inp.nlocal]).max()),
                    "mlp_ms_per_step": ph32["mlp"] / max(ph32["calls"], 1)}
            ani.set_option("mlp_arith", 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["adapter_path"] = adapter_path(inp, wl.mpath)
    wl.close()
    if rank == 0 and world == 1 and not args.no_md and not args.no_extra:
        # labelled secondary: the same MD loop with the reduced-precision opt-in (two-term fp16 splits, three MFMA products)
        k2 = min(args.steps, 100)
        dt2x, _, info2x, _ = md_pass(system, k2, min(args.warmup, 10), arith=2, measure_rebuilds=False)
        out["value_f16x2_split"] = {"value": k2 / dt2x * 0.0432, "unit": "ns/day", "ms_per_step": dt2x / k2 * 1e3, "steps": k2,
                                    "list_rebuilds": info2x["list_rebuilds"],
                                    "what": "option mlp_arith = 2 (LAMMPS_ANI_ALLOW_TF32=1): operands to 2^-22 instead of exact; NOT the "
                                            "headline; force error against the oracle in parity.f16x2_split_path"}

    if rank == 0 and world == 1 and not args.no_extra and (args.atoms, args.models) == (100002, 1):
        # BASELINE.json configs[1]: full 8-member ensemble on a ~10k-atom water box (hot path; not the headline value)
        args_models = args.models
        w2 = Workload(10002, 8)
        dt2, ph2 = w2.timed_run(args.steps, args.warmup)
        c2 = max(ph2["calls"], 1)
        out["extra_config"] = {"workload": "water-10002, ANI-2x shaped, 8 models (BASELINE.json configs[1]), hot path",
                               "ms_per_step": dt2 / args.steps * 1e3, "value": args.steps / dt2 * 0.0432, "unit": "ns/day",
                               "phase_ms": {k: ph2[k] / c2 for k in ("compact", "aev_fwd", "mlp", "aev_bwd")}}
        w2.close()
        # BASELINE.json configs[4] shape on one GPU: reactive CH4:O2 gas (3 of the 4 ANI-1x species present, ~26 list
        # entries per atom), ANI-1x shaped 8-member ensemble with the pairwise repulsion of the reactive models
        sysm = hx.combustion_box(100008, seed=12345)
        w3 = Workload(len(sysm.x), 8, repulsion=True, kind="ani1x", system=sysm)
        dt3, ph3 = w3.timed_run(args.steps, args.warmup)
        c3 = max(ph3["calls"], 1)
        out["mixed_species_config"] = {
            "workload": f"CH4:O2 1:2 gas, 0.25 g/cm3, {len(sysm.x)} atoms (H,C,O of the 4 ANI-1x species; BASELINE.json configs[4] "
                        "shape on one GPU), ANI-1x shaped, 8 models, pairwise repulsion on, hot path",
            "ms_per_step": dt3 / args.steps * 1e3, "value": args.steps / dt3 * 0.0432, "unit": "ns/day",
            "npairs_per_atom": w3.inp.npairs / max(w3.inp.nlocal, 1), "aev_columns": w3.ani.debug_view().aev_active_length,
            "phase_ms": {k: ph3[k] / c3 for k in ("compact", "aev_fwd", "mlp", "aev_bwd")}}
        w3.close()
        del args_models
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
