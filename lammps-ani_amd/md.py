"""A LAMMPS-free stand-in for the timestep loop that drives ``PairANI::compute`` in the reference's runs.

The reference is always run under LAMMPS' velocity-Verlet integrator (``run N`` with ``fix nve`` [+ ``fix langevin``],
``neighbor 2.0 bin``, ``neigh_modify every 10 delay 0 check yes`` — examples/benchmark/in.lammps:24-26,54-72); its
headline ns/day is the rate of THAT loop.  ``VerletRun`` reproduces the loop's order of operations around the C ABI with
everything resident on the GPU:

    initial_integrate -> [displacement check every `every` steps -> device neighbour list (ani_build_list_device)]
    -> forward ghost positions -> force_clear -> ani_compute_full_device -> reverse ghost forces -> post_force fixes
    -> final_integrate

Only device memory, streams and torch.distributed come from torch; the arithmetic of the hot path is libani_hip's.

Ghosts at re-neighbouring (LAMMPS: pbc + exchange + borders):
  * one rank, periodic box (``box_lo`` given): owned atoms are wrapped and the periodic-image ghosts within the
    neighbour cutoff of the box faces are regenerated on the device — the run can go on indefinitely;
  * several ranks: atoms never migrate and the ghost set chosen at set-up is kept.  The ghost shell is then taken
    ``ghost_margin`` wider than the neighbour cutoff and the run stops loudly once an owned atom has moved more than
    ``ghost_margin / 2`` from its set-up position — enough for the hundreds of steps of a benchmark or a conservation
    test, not for production MD (that is LAMMPS' job, through pair_ani.cpp).
"""
from __future__ import annotations

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
                 langevin=None, box_lo=None):
        """ani: ani_hip.ANI (full list, any precision); inp: harness.RankInput decomposed with
        ``skin = skin + ghost_margin``; langevin: None or (T_target, damp_fs) as ``fix langevin T T damp seed``."""
        from .comm import GhostExchange
        self.ani, self.device, self.group = ani, device, group
        self.nlocal, self.ntotal = inp.nlocal, inp.ntotal
        self.dt, self.cutneigh, self.skin, self.every = float(dt), cutoff + skin, skin, int(every)
        self.ghost_margin = float(ghost_margin)
        self.ex = GhostExchange(inp, box_len, device, group=group)
        self.box_len = torch.as_tensor(np.asarray(box_len, dtype=np.float64), device=device)
        # single-rank periodic mode: ghosts are regenerated at every re-neighbouring
        self.reghost = box_lo is not None and self.ex.world == 1
        if self.reghost:
            self.box_lo = torch.as_tensor(np.asarray(box_lo, dtype=np.float64), device=device)
            assert float(self.box_len.min()) >= self.cutneigh, "box shorter than the neighbour cutoff: images beyond +-1 needed"
            self._image_combos = torch.tensor([(a, b, c) for a in range(3) for b in range(3) for c in range(3) if (a, b, c) != (1, 1, 1)],
                                              dtype=torch.long, device=device)
            self._box_lo_np = np.asarray(box_lo, dtype=np.float64)
            self._box_len_np = np.asarray(box_len, dtype=np.float64)
        self.x = torch.as_tensor(inp.x, dtype=torch.float64, device=device).contiguous()
        self.species = torch.as_tensor(inp.species.astype(np.int32), device=device)
        m = torch.as_tensor(np.asarray(masses, dtype=np.float64), device=device)[self.species[: self.nlocal].long()]
        self.mass = m[:, None]
        # per-atom factors of the integrator, expanded once so that each update is ONE fused device kernel
        self._dtf_over_m = ((0.5 * float(dt) * FTM2V) / self.mass).expand(-1, 3).contiguous()
        self._lang = None
        if langevin is not None:
            T, damp = langevin
            g1 = (-self.mass / damp / FTM2V).expand(-1, 3).contiguous()
            g2 = (torch.sqrt(self.mass) * (24.0 * BOLTZ * T / damp / float(dt) / MVV2E) ** 0.5 / FTM2V).expand(-1, 3).contiguous()
            self._lang = (g1, g2, torch.empty((self.nlocal, 3), dtype=torch.float64, device=device))
        self.v = torch.zeros((self.nlocal, 3), dtype=torch.float64, device=device)
        self.f = torch.zeros((self.ntotal, 3), dtype=torch.float64, device=device)
        self.ev = torch.zeros(10, dtype=torch.float64, device=device)
        self.x_setup = self.x[: self.nlocal].clone()
        self.x_built = self.x[: self.nlocal].clone()
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + 7919 * (dist.get_rank(group) if dist.is_initialized() else 0))
        self.langevin = langevin
        self.step_no = 0
        self.since_build = 0
        self.nbuilds = 0
        self.npairs = 0
        self._stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else None
        self._build_list()
        self._forces()

    # ---- pieces of the loop ---------------------------------------------------------------------------
    def _allreduce_max(self, t: torch.Tensor) -> float:
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and t.is_cuda:
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
                return float(c)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t)

    def _regenerate_ghosts(self):
        """Domain::pbc() + Comm::borders() for one rank owning the whole periodic box."""
        n, L, lo, cut = self.nlocal, self.box_len, self.box_lo, self.cutneigh
        xl = self.x[:n]
        xl -= torch.floor((xl - lo) / L) * L
        near_lo, near_hi = xl < lo + cut, xl >= lo + L - cut
        # the 26 image shifts at once: cond[s + 1, atom, d] says whether the atom has an image displaced by s box lengths
        # along d; one nonzero() (one host sync) lists the (shift, atom) pairs shift-major, atoms ascending
        cond = torch.stack([near_hi, torch.ones_like(near_lo), near_lo], 0)
        cb = self._image_combos
        mask = cond[cb[:, 0], :, 0] & cond[cb[:, 1], :, 1] & cond[cb[:, 2], :, 2]
        hit = mask.nonzero()
        owner = hit[:, 1]
        shift = (cb[hit[:, 0]] - 1).to(torch.float64) * L
        self.ex.reset_single(owner, shift)
        self.ntotal = n + int(owner.numel())
        self.x = torch.cat([xl, xl[owner] + shift]).contiguous()
        self.species = torch.cat([self.species[:n], self.species[:n][owner]]).contiguous()
        self.f = torch.zeros((self.ntotal, 3), dtype=torch.float64, device=self.device)

    def _build_list(self):
        """Neighbor::build's role: ghosts refreshed, then the full list on the device."""
        if self.reghost:
            self._regenerate_ghosts()
        self.ex.forward_positions(self.x)
        moved = (self.x[: self.nlocal] - self.x_setup).square().sum(1).max() if self.nlocal else torch.zeros((), device=self.device)
        moved = self._allreduce_max(moved.reshape(1).clone()) ** 0.5
        if self.nbuilds and not self.reghost and moved > 0.5 * self.ghost_margin:
            raise RuntimeError(f"an atom moved {moved:.2f} A from its set-up position, more than ghost_margin/2 = "
                               f"{0.5 * self.ghost_margin:.2f} A: the fixed ghost shell of this stand-in no longer "
                               "covers the neighbour cutoff (re-decompose, or raise ghost_margin)")
        if self.reghost:   # owned atoms were just wrapped into the box and the ghosts lie within cutneigh of its faces
            lo = self._box_lo_np - self.cutneigh - 0.25
            hi = self._box_lo_np + self._box_len_np + self.cutneigh + 0.25
        else:
            lo = (self.x.min(0).values - 0.25).cpu().numpy()
            hi = (self.x.max(0).values + 0.25).cpu().numpy()
        self.npairs = self.ani.build_list_device(self.ntotal, self.nlocal, self.species.data_ptr(), self.x.data_ptr(),
                                                 self.cutneigh, lo, hi, stream=self._stream)
        self.x_built.copy_(self.x[: self.nlocal])
        self.since_build = 0
        self.nbuilds += 1

    def _forces(self):
        self.f.zero_()
        self.ani.compute_device(self.ntotal, self.nlocal, None, self.x.data_ptr(), self.npairs, None, None, None, 1,
                                self.f.data_ptr(), self.ev.data_ptr(), stream=self._stream)
        self.ex.reverse_add(self.f)
        if self._lang is not None:
            # fix langevin (LAMMPS fix_langevin.cpp, uniform random numbers in [-0.5, 0.5)): f += g1 v + g2 r on owned atoms
            g1, g2, r = self._lang
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
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and t.is_cuda:
                c = t.cpu()
                dist.all_reduce(c, group=self.group)
                return float(c)
            dist.all_reduce(t, group=self.group)
        return float(t)

    def step(self):
        # fix nve initial_integrate: v += dtf f / m ; x += dt v
        self.v.addcmul_(self.f[: self.nlocal], self._dtf_over_m)
        self.x[: self.nlocal].add_(self.v, alpha=self.dt)
        self.step_no += 1
        self.since_build += 1
        # Neighbor::decide + check_distance (every N steps, rebuild if any atom moved more than skin/2)
        rebuild = False
        if self.since_build % self.every == 0:
            d2 = (self.x[: self.nlocal] - self.x_built).square().sum(1).max().reshape(1) if self.nlocal else \
                torch.zeros(1, dtype=torch.float64, device=self.device)
            rebuild = self._allreduce_max(d2.clone()) > (0.5 * self.skin) ** 2
        if rebuild:
            self._build_list()
        else:
            self.ex.forward_positions(self.x)
        self._forces()
        # fix nve final_integrate
        self.v.addcmul_(self.f[: self.nlocal], self._dtf_over_m)

    # ---- thermo ------------------------------------------------------------------------------------------
    def kinetic_energy(self) -> float:
        ke = 0.5 * MVV2E * (self.mass * self.v.square()).sum().reshape(1)
        return self._allreduce_sum(ke.clone())

    def potential_energy(self) -> float:
        return self._allreduce_sum(self.ev[:1].clone())
