"""A LAMMPS-free stand-in for the timestep loop that drives ``PairANI::compute`` in the reference's runs.

The reference is always run under LAMMPS' velocity-Verlet integrator (``run N`` with ``fix nve`` [+ ``fix langevin``],
``neighbor 2.0 bin``, ``neigh_modify every 10 delay 0 check yes`` — examples/benchmark/in.lammps:24-26,54-72); its
headline ns/day is the rate of THAT loop.  ``VerletRun`` reproduces the loop's order of operations around the C ABI with
everything resident on the GPU:

    initial_integrate -> [displacement check every `every` steps -> device neighbour list (ani_build_list_device)]
    -> forward ghost positions -> force_clear -> ani_compute_full_device -> reverse ghost forces -> post_force fixes
    -> final_integrate

Only device memory, streams and torch.distributed come from torch; the arithmetic of the hot path is libani_hip's.

Re-neighbouring (LAMMPS: Domain::pbc + Comm::exchange + Comm::borders + Neighbor::build) is done on the device as
well, by ``comm.DomainComm``: owned atoms are wrapped into the box and handed to the rank whose brick now holds them,
the ghost shell (periodic images included) is rebuilt from the current positions, then the library builds the full
list (ani_build_list_device).  One rank or several, the run can go on indefinitely; between rebuilds only the ghost
positions (forward) and ghost forces (reverse) travel, one message per peer and direction.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

# LAMMPS `units real` constants (update.cpp of LAMMPS): forces kcal/mol/A, masses g/mol, time fs
FTM2V = 1.0 / 48.88821291 / 48.88821291
MVV2E = 48.88821291 * 48.88821291
BOLTZ = 0.0019872067

# g/mol by ANI-2x species order H C N O S F Cl (the reference's data files carry the same values in `Masses`)
ANI2X_MASSES = (1.008, 12.011, 14.007, 15.999, 32.06, 18.998, 35.45)


class VerletRun:
    def __init__(self, ani, inp, box_len, device, dt: float = 0.5, cutoff: float = 5.1, skin: float = 2.0,
                 ghost_margin: float = 0.0, every: int = 10, masses=ANI2X_MASSES, group=None, seed: int = 12345,
                 langevin=None, box_lo=None, grid=None, periodic=(True, True, True), overlap=None, native_comm=None,
                 force_collectives=False, vflag=False):
        """ani: ani_hip.ANI (full list, any precision); inp: harness.RankInput of this rank — only its OWNED atoms
        (positions, types, global tags) are taken, ghosts and lists are rebuilt here; langevin: None or
        (T_target, damp_fs) as ``fix langevin T T damp seed``; grid: processor grid (default comm.grid_for(world));
        ghost_margin: ignored (kept for callers of the earlier fixed-ghost-shell version); overlap: run the two ghost
        exchanges of a step on a second stream beside the rows that do not need them (the library's split step,
        include/ani_hip.h ani_step_*).  Cutting the step costs five more launches (+0.05 ms at 12 500 atoms per GPU,
        measured on one card), which the hidden exchanges have to pay back: default on with several ranks over RCCL (two
        latency-bound all-to-alls per step), off otherwise; environment ANI_MD_OVERLAP=0/1 overrides the default.
        native_comm: an ani_hip.NativeComm -- the exchanges then run inside libani_hip.so as grouped ncclSend / ncclRecv
        (include/ani_comm.h) instead of torch.distributed collectives; force_collectives: a single rank takes the several-rank
        paths (one-participant collectives), so that one GPU can exercise them.  vflag: every force evaluation also
        accumulates the pair style's virial (``virial()``; LAMMPS sets vflag on thermo steps with a pressure compute)."""
        from .comm import DomainComm, grid_for
        self.ani, self.device, self.group = ani, device, group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dt, self.cutneigh, self.skin, self.every = float(dt), cutoff + skin, skin, int(every)
        box_lo = np.zeros(3) if box_lo is None else np.asarray(box_lo, dtype=np.float64)
        self._box_lo_np = box_lo
        self._box_len_np = np.asarray(box_len, dtype=np.float64)
        self.dc = DomainComm(grid or grid_for(world), box_lo, box_len, self.cutneigh, device, group=group, periodic=periodic,
                             native=native_comm, force_collectives=force_collectives)
        self.native = native_comm
        self.vflag = bool(vflag)
        self.ex = self.dc   # the exchange object (forward_positions / reverse_add)
        self.masses = torch.as_tensor(np.asarray(masses, dtype=np.float64), device=device)
        self.langevin = langevin
        n = inp.nlocal
        self.nlocal = n
        self.x = torch.as_tensor(inp.x[:n], dtype=torch.float64, device=device).contiguous()
        self.species = torch.as_tensor(inp.species[:n].astype(np.int32), device=device)
        self.tag = torch.as_tensor(np.asarray(inp.tag[:n]).astype(np.int64), device=device)
        self.v = torch.zeros((n, 3), dtype=torch.float64, device=device)
        self.ev = torch.zeros(10, dtype=torch.float64, device=device)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + 7919 * (dist.get_rank(group) if dist.is_initialized() else 0))
        self.step_no = 0
        self.since_build = 0
        self.nbuilds = 0
        self.npairs = 0
        self._stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else None
        # on the GPU the loop's own steps (fix nve, fix langevin, displacement check, one-rank ghost copies) are the fused
        # kernels of include/ani_md.h; the tensor-operation forms below remain for reading and for other devices
        self._fused = torch.device(device).type == "cuda"
        self._seed = int(seed)
        if self._fused:
            from . import ani_hip
            self._md = ani_hip.lib()
            self._d2max = torch.zeros(1, dtype=torch.float64, device=device)
            self._sendbuf = torch.empty((0, 3), dtype=torch.float64, device=device)
            self._recvbuf = torch.empty((0, 3), dtype=torch.float64, device=device)
            ani.set_option("device_overwrite_forces", 1)   # no separate force_clear launch
        env = os.environ.get("ANI_MD_OVERLAP")
        # default off: cutting the step costs +0.05 ms on one card and its benefit has not been measured on several
        want = False if overlap is None else bool(overlap)
        if env is not None and overlap is None:
            want = env not in ("", "0")
        self._overlap = bool(want and self._fused)
        self._want_fold = os.environ.get("ANI_MD_FOLD", "1") not in ("", "0")   # measurement knob: 0 keeps the two ghost kernels
        self._fold = False
        # one rank on the GPU: re-neighbouring (position wrap, ghost shell, buffers, displacement check) through the kernels of
        # include/ani_md.h instead of tensor operations; ANI_MD_NATIVE_REBUILD=0 keeps the tensor forms (measurements, tests)
        self._native_rebuild = bool(self._fused and not self.dc.multi and
                                    os.environ.get("ANI_MD_NATIVE_REBUILD", "1") not in ("", "0"))
        self._cap = 0
        if self._overlap:
            # ANI_MD_OVERLAP_ONE_STREAM: measurement knob, the same cut step with everything on the compute stream
            self._comm_stream = torch.cuda.current_stream(device) if os.environ.get("ANI_MD_OVERLAP_ONE_STREAM") else \
                torch.cuda.Stream(device=device)
            self._ev = [torch.cuda.Event() for _ in range(4)]
        self._build_list()
        self._forces()
        self._post_force()

    # ---- pieces of the loop ---------------------------------------------------------------------------
    def _allreduce_max(self, t: torch.Tensor) -> float:
        if self.native is not None and self.dc.multi:
            self.native.allreduce(t.data_ptr(), t.numel(), "max", stream=self._stream)
            return float(t)
        if dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.dc.multi):
            if dist.get_backend(self.group) == "gloo" and t.is_cuda:
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
                return float(c)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t)

    def _per_atom_factors(self):
        """mass-dependent factors of the integrator, expanded once per re-neighbouring (atoms may have migrated) so
        that each update is ONE fused device kernel"""
        n = self.nlocal
        if not self.dc.multi and getattr(self, "_factors_n", -1) == n:
            return   # one rank: nobody migrates, the owned atoms keep their order
        self._factors_n = n
        m = self.masses[self.species[:n].long()]
        self.mass = m[:, None]
        self._dtfm1 = (0.5 * self.dt * FTM2V) / m                        # [n], the fused kernels' form
        self._lang = None
        self._g1 = self._g2 = None
        if self.langevin is not None:
            T, damp = self.langevin
            self._g1 = m * (-1.0 / damp / FTM2V)
            self._g2 = torch.sqrt(m) * ((24.0 * BOLTZ * T / damp / self.dt / MVV2E) ** 0.5 / FTM2V)
        if self._fused:
            return   # the [n, 3] forms below are the operands of the tensor-operation steps only
        self._dtf_over_m = self._dtfm1[:, None].expand(-1, 3).contiguous()
        if self.langevin is not None:
            self._lang = (self._g1[:, None].expand(-1, 3).contiguous(), self._g2[:, None].expand(-1, 3).contiguous(),
                          torch.empty((n, 3), dtype=torch.float64, device=self.device))

    def _grow(self, ntotal: int):
        """capacity buffers of the per-atom arrays that include ghosts (grown 1.5x, owned rows kept)"""
        if ntotal <= self._cap:
            return
        cap = ntotal + ntotal // 2 + 64
        n = self.nlocal
        xb = torch.empty((cap, 3), dtype=torch.float64, device=self.device)
        sb = torch.empty(cap, dtype=torch.int32, device=self.device)
        xb[:n].copy_(self.x[:n])
        sb[:n].copy_(self.species[:n])
        self._xbuf, self._spbuf = xb, sb
        self._fbuf = torch.zeros((cap, 3), dtype=torch.float64, device=self.device)
        self._sidx = torch.empty(cap, dtype=torch.int64, device=self.device)
        self._sshift = torch.empty((cap, 3), dtype=torch.float64, device=self.device)
        self._cap = cap

    def _build_list_native(self):
        """Domain::pbc + Comm::borders + Neighbor::build of a one-rank run without tensor operations: wrap kernel, ghost shell by
        count / scan / fill (one host read of the ghost count), ghosts appended in place, device list, ghost fold."""
        n, dc, md, st = self.nlocal, self.dc, self._md, self._stream
        if self._cap == 0:
            self._grow(max(self.x.shape[0], n))
            self._xbuilt_buf = torch.empty((n, 3), dtype=torch.float64, device=self.device)
            self._chk_dev = torch.zeros(1, dtype=torch.float64, device=self.device)
            self._chk_host = torch.zeros(1, dtype=torch.float64).pin_memory()
            self._clo2 = dc._clo.reshape(-1, 3).contiguous()
            self._chi2 = dc._chi.reshape(-1, 3).contiguous()
            self._cshift2 = dc._shift.reshape(-1, 3).contiguous()
            self._ncombo = int(self._clo2.shape[0])
            nblk = max((n + 255) // 256, 1)
            self._blk = torch.empty(2 * max(self._ncombo, 1) * nblk, dtype=torch.int32, device=self.device)
            self._cnt_dev = torch.zeros(self._ncombo + 1, dtype=torch.int32, device=self.device)
            self._cnt_host = torch.zeros(self._ncombo + 1, dtype=torch.int32).pin_memory()
            self._lo3 = np.ascontiguousarray(self._box_lo_np, dtype=np.float64)
            self._len3 = np.ascontiguousarray(self._box_len_np, dtype=np.float64)
            self._pmask = sum(1 << k for k in range(3) if dc.periodic[k])
        x = self._xbuf
        self._check(md.ani_md_wrap_positions(x.data_ptr(), n, self._lo3.ctypes.data, self._len3.ctypes.data, self._pmask, st))
        ng = 0
        nblk = max((n + 255) // 256, 1)
        half = max(self._ncombo, 1) * nblk
        if self._ncombo:
            self._check(md.ani_md_ghost_shell_count(x.data_ptr(), n, self._clo2.data_ptr(), self._chi2.data_ptr(), self._ncombo,
                                                    self._blk.data_ptr(), self._blk[half:].data_ptr(), self._cnt_dev.data_ptr(), st))
            self._cnt_host.copy_(self._cnt_dev, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()      # the ghost count sizes the buffers (rebuild steps only)
            ng = int(self._cnt_host[0])
        if n + ng > self._cap:
            self.x, self.species = self._xbuf[:n], self._spbuf[:n]
            self._grow(n + ng)
            x = self._xbuf
        if ng:
            self._check(md.ani_md_ghost_shell_fill(x.data_ptr(), n, self._clo2.data_ptr(), self._chi2.data_ptr(), self._cshift2.data_ptr(),
                                                   self._ncombo, self._blk[half:].data_ptr(), self._sidx.data_ptr(),
                                                   self._sshift.data_ptr(), st))
            self._check(md.ani_md_append_ghosts(x.data_ptr(), self._spbuf.data_ptr(), n, self._sidx.data_ptr(), self._sshift.data_ptr(), ng, st))
        dc.send_idx, dc.send_shift = self._sidx[:ng], self._sshift[:ng]
        dc.send_splits, dc.recv_splits, dc.nlocal, dc.nghost = [ng], [ng], n, ng
        self.ntotal = n + ng
        self.x, self.species, self.f = self._xbuf[: self.ntotal], self._spbuf[: self.ntotal], self._fbuf[: self.ntotal]
        self._per_atom_factors()
        lo = dc.sub_lo - self.cutneigh - 0.25
        hi = dc.sub_hi + self.cutneigh + 0.25
        # the fold goes in with the list: its check rides on the build's own synchronisation
        self._fold = bool(self._want_fold and self.ani.use_single and not self._overlap)
        if self._fold:
            self.ani.stage_ghost_fold(dc.send_idx.data_ptr(), dc.send_shift.data_ptr(), ng)
        self.npairs = self.ani.build_list_device(self.ntotal, n, self.species.data_ptr(), self.x.data_ptr(), self.cutneigh, lo, hi, stream=st)
        self._xbuilt_buf.copy_(self.x[:n])
        self.x_built = self._xbuilt_buf
        self._check(md.ani_md_check(self._d2max.data_ptr(), self.ev.data_ptr(), self._chk_dev.data_ptr(), st))   # zeroes the maximum
        self.since_build = 0
        self.nbuilds += 1

    def _build_list(self):
        """Domain::pbc + Comm::exchange + Comm::borders + Neighbor::build."""
        if self._native_rebuild:
            return self._build_list_native()
        n = self.nlocal
        xo, v, tag, sp = self.dc.exchange(self.x[:n], self.v, self.tag, self.species[:n])
        self.nlocal = n = xo.shape[0]
        self.v, self.tag = v.contiguous(), tag.contiguous()
        self.x, self.species = self.dc.borders(xo, sp.contiguous())
        self.ntotal = self.x.shape[0]
        self._per_atom_factors()
        self.f = torch.zeros((self.ntotal, 3), dtype=torch.float64, device=self.device)
        # every atom lies inside the rank's brick widened by the ghost cutoff
        lo = self.dc.sub_lo - self.cutneigh - 0.25
        hi = self.dc.sub_hi + self.cutneigh + 0.25
        self.npairs = self.ani.build_list_device(self.ntotal, n, self.species.data_ptr(), self.x.data_ptr(),
                                                 self.cutneigh, lo, hi, stream=self._stream)
        # one rank: every ghost is an image of an owned atom -- the library's first and last kernel of a step do the two ghost
        # exchanges themselves (ani_set_ghost_fold); the maps belong to this list
        self._fold = False
        if self._fused and self._want_fold and not self.dc.multi and self.ani.use_single and not self._overlap:
            self.ani.set_ghost_fold(self.dc.send_idx.data_ptr(), self.dc.send_shift.data_ptr(), self.ntotal - n, stream=self._stream)
            self._fold = True
        self.x_built = self.x[:n].clone()
        if self._fused:
            self._d2max.zero_()
        self.since_build = 0
        self.nbuilds += 1

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"ani_md kernel launch failed (hipError {rc})")

    def _forces(self):
        """force_clear + PairANI::compute + reverse communication of the ghost forces"""
        if not self._fused:
            self.f.zero_()
        self.ani.compute_device(self.ntotal, self.nlocal, None, self.x.data_ptr(), self.npairs, None, None, None, 1,
                                self.f.data_ptr(), self.ev.data_ptr(), vflag=self.vflag, stream=self._stream)
        if self._fold:
            pass   # the finish kernel has added the images' rows into their owners'
        elif self._fused and not self.dc.multi:
            self._check(self._md.ani_md_reverse_ghosts(self.f.data_ptr(), self.dc.send_idx.data_ptr(), self.nlocal,
                                                       self.ntotal - self.nlocal, self._stream))
        elif self._fused and self.native is not None:
            self.native.reverse(self.f.data_ptr(), self.nlocal, stream=self._stream)
        elif self._fused:
            dc, ns = self.dc, int(self.dc.send_idx.numel())
            if self._recvbuf.shape[0] != ns:
                self._recvbuf = torch.empty((ns, 3), dtype=torch.float64, device=self.device)
            dc._a2a(self.f[self.nlocal:], dc.send_splits, dc.recv_splits, out=self._recvbuf)
            self._check(self._md.ani_md_unpack_reverse(self.f.data_ptr(), dc.send_idx.data_ptr(), ns, self._recvbuf.data_ptr(),
                                                       self._stream))
        else:
            self.dc.reverse_add(self.f)

    def _post_force(self):
        """fix langevin (LAMMPS fix_langevin.cpp, uniform random numbers in [-0.5, 0.5)): f += g1 v + g2 r on owned atoms.
        Tensor-operation form: set-up, and devices without the fused kernels (step() fuses it with final_integrate)."""
        if self._g1 is not None:
            if self._lang is not None:
                g1, g2, r = self._lang
            else:   # fused mode keeps only the [n] factors: this form runs once, at set-up
                g1, g2 = self._g1[:, None], self._g2[:, None]
                r = torch.empty((self.nlocal, 3), dtype=torch.float64, device=self.device)
            r.uniform_(-0.5, 0.5, generator=self.gen)
            fl = self.f[: self.nlocal]
            fl.addcmul_(g1, self.v)
            fl.addcmul_(g2, r)

    def create_velocities(self, T: float):
        """``velocity all create T seed mom yes dist gaussian`` (examples/benchmark/in.lammps:54)."""
        if T <= 0.0:
            self.v.zero_()
            return
        sigma = torch.sqrt(BOLTZ * T / (self.mass * MVV2E))
        self.v = sigma * torch.randn((self.nlocal, 3), dtype=torch.float64, device=self.device, generator=self.gen)
        p = (self.mass * self.v).sum(0)
        mtot = self.mass.sum().reshape(1)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            pm = torch.cat([p, mtot])
            pm = pm.cpu() if dist.get_backend(self.group) == "gloo" else pm
            dist.all_reduce(pm, group=self.group)
            pm = pm.to(self.device)
            p, mtot = pm[:3], pm[3:]
        self.v -= p / mtot
        ke = self.kinetic_energy()
        n = self._allreduce_sum(torch.tensor([float(self.nlocal)], dtype=torch.float64, device=self.device))
        t_now = 2.0 * ke / (3.0 * n - 3.0) / BOLTZ
        self.v *= (T / t_now) ** 0.5

    def _allreduce_sum(self, t: torch.Tensor) -> float:
        if self.native is not None and self.dc.multi and t.is_cuda and t.dtype == torch.float64:
            self.native.allreduce(t.data_ptr(), t.numel(), "sum", stream=self._stream)
            return float(t)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and t.is_cuda:
                c = t.cpu()
                dist.all_reduce(c, group=self.group)
                return float(c)
            dist.all_reduce(t, group=self.group)
        return float(t)

    def _initial_integrate(self):
        # fix nve initial_integrate: v += dtf f / m ; x += dt v  (fused: + the displacement maximum of check_distance)
        if self._fused:
            self._check(self._md.ani_md_initial_integrate(self.x.data_ptr(), self.v.data_ptr(), self.f.data_ptr(),
                                                          self._dtfm1.data_ptr(), self.dt, self.nlocal, self.x_built.data_ptr(),
                                                          self._d2max.data_ptr(), self._stream))
        else:
            self.v.addcmul_(self.f[: self.nlocal], self._dtf_over_m)
            self.x[: self.nlocal].add_(self.v, alpha=self.dt)

    def _middle(self, force_rebuild: bool = False):
        """between the two integrator halves: re-neighbouring decision, ghost positions, forces, ghost forces"""
        self.step_no += 1
        self.since_build += 1
        # Neighbor::decide + check_distance (every N steps, rebuild if any atom moved more than skin/2)
        rebuild = bool(force_rebuild)
        if not rebuild and self.since_build % self.every == 0:
            if self._native_rebuild:
                # one kernel and one pinned read: the running maximum (or +inf when the last energy is not finite), then zeroed
                self._check(self._md.ani_md_check(self._d2max.data_ptr(), self.ev.data_ptr(), self._chk_dev.data_ptr(), self._stream))
                self._chk_host.copy_(self._chk_dev, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
                worst = float(self._chk_host[0])
                if worst == float("inf"):
                    raise RuntimeError("non-finite energy from the device step: a neighbour count exceeded the kernels' LDS "
                                       "capacity (ANI_ERR_CAPACITY) or the forces diverged")
                rebuild = worst > (0.5 * self.skin) ** 2
                d2 = None
            elif self._fused:
                d2 = self._d2max.clone()    # running maximum since the last look (monotone between rebuilds: same decision)
                self._d2max.zero_()
            else:
                d2 = (self.x[: self.nlocal] - self.x_built).square().sum(1).max().reshape(1) if self.nlocal else \
                    torch.zeros(1, dtype=torch.float64, device=self.device)
            if d2 is not None:
                # the same host round trip carries the health of the last force evaluation: the device entry point cannot
                # return ANI_ERR_CAPACITY (nothing synchronises), it turns the energy into NaN instead
                d2 = torch.where(torch.isfinite(self.ev[:1]), d2, torch.full_like(d2, float("inf")))
                worst = self._allreduce_max(d2.clone())
                if worst == float("inf"):
                    raise RuntimeError("non-finite energy from the device step: a neighbour count exceeded the kernels' LDS "
                                       "capacity (ANI_ERR_CAPACITY) or the forces diverged")
                rebuild = worst > (0.5 * self.skin) ** 2
        if rebuild:
            self._build_list()
        elif self._overlap:
            self._forces_overlapped()
            return
        elif self._fold:
            pass   # the pack kernel of the step reads the images' positions from their owners (and writes them to x)
        elif self._fused and not self.dc.multi:
            self._check(self._md.ani_md_forward_ghosts(self.x.data_ptr(), self.dc.send_idx.data_ptr(), self.dc.send_shift.data_ptr(),
                                                       self.nlocal, self.ntotal - self.nlocal, self._stream))
        elif self._fused and self.native is not None:
            self.native.forward(self.x.data_ptr(), self.nlocal, stream=self._stream)
        elif self._fused:
            # several ranks: one pack kernel, the all-to-all receives straight into the ghost block of x
            dc, ns = self.dc, int(self.dc.send_idx.numel())
            if self._sendbuf.shape[0] != ns:
                self._sendbuf = torch.empty((ns, 3), dtype=torch.float64, device=self.device)
            self._check(self._md.ani_md_pack_ghosts(self.x.data_ptr(), dc.send_idx.data_ptr(), dc.send_shift.data_ptr(), ns,
                                                    self._sendbuf.data_ptr(), self._stream))
            dc._a2a(self._sendbuf, dc.recv_splits, dc.send_splits, out=self.x[self.nlocal:])
        else:
            self.dc.forward_positions(self.x)
        self._forces()

    def step(self, force_rebuild: bool = False):
        """force_rebuild: re-neighbour in this step whatever the displacement check would say (measurements)"""
        self._initial_integrate()
        self._middle(force_rebuild)
        self._final_integrate()

    def run(self, nsteps: int, before_forces=None, after_forces=None):
        """``run N`` without per-step output: nothing looks at the full-step velocities between two steps, so the
        final_integrate of a step and the initial_integrate of the next are ONE kernel (same arithmetic and rounding as
        step() N times; the loop is in a full-step state again when this returns).  before_forces(k) / after_forces(k):
        optional callables around the force evaluation of step k (the bench's phase-timer switches)."""
        if nsteps <= 0:
            return
        if not self._fused:
            for _ in range(nsteps):
                self.step()
            return
        self._initial_integrate()
        for k in range(nsteps):
            if before_forces is not None:
                before_forces(k)
            self._middle()
            if after_forces is not None:
                after_forces(k)
            if k + 1 == nsteps:
                self._final_integrate()
            else:
                lang = self._g1 is not None
                self._check(self._md.ani_md_final_initial_integrate(
                    self.x.data_ptr(), self.v.data_ptr(), self.f.data_ptr(), self._dtfm1.data_ptr(), self.dt, self.nlocal,
                    1 if lang else 0, self._g1.data_ptr() if lang else None, self._g2.data_ptr() if lang else None,
                    self.tag.data_ptr(), self._seed, self.step_no, self.x_built.data_ptr(), self._d2max.data_ptr(), self._stream))

    def _pack_and_send_ghosts(self):
        """forward exchange on the current stream: x[nlocal:] <- the owners' positions (+ image shifts)"""
        dc = self.dc
        if not dc.multi:
            self._check(self._md.ani_md_forward_ghosts(self.x.data_ptr(), dc.send_idx.data_ptr(), dc.send_shift.data_ptr(),
                                                       self.nlocal, self.ntotal - self.nlocal, torch.cuda.current_stream(self.device).cuda_stream))
            return
        if self.native is not None:
            self.native.forward(self.x.data_ptr(), self.nlocal, stream=torch.cuda.current_stream(self.device).cuda_stream)
            return
        ns = int(dc.send_idx.numel())
        if self._sendbuf.shape[0] != ns:
            self._sendbuf = torch.empty((ns, 3), dtype=torch.float64, device=self.device)
        self._check(self._md.ani_md_pack_ghosts(self.x.data_ptr(), dc.send_idx.data_ptr(), dc.send_shift.data_ptr(), ns,
                                                self._sendbuf.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        dc._a2a(self._sendbuf, dc.recv_splits, dc.send_splits, out=self.x[self.nlocal:])

    def _forces_overlapped(self):
        """Forward exchange, PairANI::compute and reverse exchange of a step that keeps its list, the two exchanges on the
        communication stream: the forward one beside the rows without ghosts (pack, compaction, AEV forward), the reverse
        one beside their backward pass (ani_step_begin / ani_step_ghosts_ready / ani_step_finish)."""
        s1, s2, ev, dc = torch.cuda.current_stream(self.device), self._comm_stream, self._ev, self.dc
        nl, nt = self.nlocal, self.ntotal
        ev[0].record(s1)                                   # initial_integrate has moved the owned atoms
        with torch.cuda.stream(s2):
            s2.wait_event(ev[0])
            self._pack_and_send_ghosts()
            ev[1].record(s2)
        self.ani.step_begin(nt, nl, self.x.data_ptr(), self.f.data_ptr(), self.ev.data_ptr(), stream=s1.cuda_stream)
        s1.wait_event(ev[1])                               # ghost positions are in place
        self.ani.step_ghosts_ready(stream=s1.cuda_stream)
        ev[2].record(s1)                                   # ghost rows of f are final
        with torch.cuda.stream(s2):
            s2.wait_event(ev[2])
            if dc.multi and self.native is not None:
                self.native.reverse_send(self.f.data_ptr(), nl, stream=s2.cuda_stream)
            elif dc.multi:
                ns = int(dc.send_idx.numel())
                if self._recvbuf.shape[0] != ns:
                    self._recvbuf = torch.empty((ns, 3), dtype=torch.float64, device=self.device)
                dc._a2a(self.f[nl:], dc.send_splits, dc.recv_splits, out=self._recvbuf)
            ev[3].record(s2)
        self.ani.step_finish(stream=s1.cuda_stream)
        s1.wait_event(ev[3])
        if dc.multi and self.native is not None:
            self.native.reverse_unpack(self.f.data_ptr(), stream=s1.cuda_stream)
        elif dc.multi:
            self._check(self._md.ani_md_unpack_reverse(self.f.data_ptr(), dc.send_idx.data_ptr(), int(dc.send_idx.numel()),
                                                       self._recvbuf.data_ptr(), s1.cuda_stream))
        else:   # one rank: the owners are here, the ghost rows are added once the owned rows are written
            self._check(self._md.ani_md_reverse_ghosts(self.f.data_ptr(), dc.send_idx.data_ptr(), nl, nt - nl, s1.cuda_stream))

    def _final_integrate(self):
        # fix langevin post_force + fix nve final_integrate
        if self._fused:
            lang = self._g1 is not None
            self._check(self._md.ani_md_final_integrate(self.v.data_ptr(), self.f.data_ptr(), self._dtfm1.data_ptr(), self.nlocal,
                                                        1 if lang else 0, self._g1.data_ptr() if lang else None,
                                                        self._g2.data_ptr() if lang else None, self.tag.data_ptr(), self._seed,
                                                        self.step_no, self._stream))
        else:
            self._post_force()
            self.v.addcmul_(self.f[: self.nlocal], self._dtf_over_m)

    def warm_paths(self):
        """Run once, without touching the trajectory, the pieces of the loop that only some steps execute — the displacement
        check with its host round trip — so that their first-use costs (module loads of the tensor kernels, lazily made
        buffers) fall before a caller's timed region whatever its number of warm-up steps."""
        if self._native_rebuild:
            scratch = torch.zeros(1, dtype=torch.float64, device=self.device)   # not the loop's own maximum: the kernel zeroes it
            self._check(self._md.ani_md_check(scratch.data_ptr(), self.ev.data_ptr(), self._chk_dev.data_ptr(), self._stream))
            self._chk_host.copy_(self._chk_dev, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            return
        if self._fused:
            d2 = self._d2max.clone()
        else:
            d2 = (self.x[: self.nlocal] - self.x_built).square().sum(1).max().reshape(1) if self.nlocal else \
                torch.zeros(1, dtype=torch.float64, device=self.device)
        d2 = torch.where(torch.isfinite(self.ev[:1]), d2, torch.full_like(d2, float("inf")))
        self._allreduce_max(d2.clone())

    # ---- thermo ------------------------------------------------------------------------------------------
    def kinetic_energy(self) -> float:
        ke = 0.5 * MVV2E * (self.mass * self.v.square()).sum().reshape(1)
        return self._allreduce_sum(ke.clone())

    def potential_energy(self) -> float:
        return self._allreduce_sum(self.ev[:1].clone())

    def virial(self) -> np.ndarray:
        """the pair style's virial of the last force evaluation, summed over ranks (kcal/mol, row-major 3x3; needs vflag)"""
        if not self.vflag:
            raise RuntimeError("VerletRun was made without vflag")
        w = self.ev[1:10].clone()
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and w.is_cuda:
                c = w.cpu()
                dist.all_reduce(c, group=self.group)
                w = c
            else:
                dist.all_reduce(w, group=self.group)
        return w.cpu().numpy().reshape(3, 3)

    def temperature(self, natoms_all: int) -> float:
        """LAMMPS compute temp: 2 KE / ((3 N - 3) k_B)"""
        return 2.0 * self.kinetic_energy() / ((3.0 * natoms_all - 3.0) * BOLTZ)
