"""Ghost-atom exchange between ranks over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" on CPU).

Replaces, for the LAMMPS-free harness, what LAMMPS' Comm does around the pair style:
  * reverse: ghost forces summed into their owners — ``comm->reverse_comm(this)`` with
    ``pack_reverse_comm`` / ``unpack_reverse_comm`` (src/pair_ani.cpp:197-201,461-484), 3 values per ghost;
  * forward: owners' positions copied to their ghosts every step (LAMMPS Verlet ``comm->forward_comm()``).
The reference's LAMMPS does 6 dependent face swaps on the host; here every rank sends each peer ONE message per
direction (a single all_to_all_single with per-peer splits), which is what a latency-bound ~50 KB exchange
wants on point-to-point xGMI links.  Periodic self-images are the rank's own chunk of the same collective.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class GhostExchange:
    def __init__(self, inp, box_len, device, dtype=torch.float64, group=None):
        """inp: harness.RankInput of this rank; box_len: [3] box lengths (for image shifts of forwarded positions)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nlocal, self.nghost = inp.nlocal, inp.nghost
        self.device, self.dtype = device, dtype
        # rehearsal mode: gloo cannot move device tensors, so messages are staged through the host (only used to run
        # the multi-rank path on a single GPU; the real path is backend nccl = RCCL with device buffers)
        self.identity_perm = False
        self.host_staged = (self.world > 1 and dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda")
        order = np.argsort(inp.owner_rank, kind="stable")
        self.ghost_perm = torch.as_tensor(order.astype(np.int64), device=device)          # ghosts sorted by owner
        send_counts = np.bincount(inp.owner_rank, minlength=self.world).astype(np.int64)
        self.send_splits = send_counts.tolist()
        lidx_sorted = torch.as_tensor(inp.owner_lidx[order].astype(np.int64), device=device)
        self.shift = torch.as_tensor((inp.shift[order] * np.asarray(box_len)[None, :]), device=device, dtype=dtype)
        if self.world > 1:
            cdev = "cpu" if self.host_staged else device
            sc = torch.as_tensor(send_counts, device=cdev)
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=group)
            self.recv_splits = rc.cpu().tolist()
            ridx = torch.empty(int(sum(self.recv_splits)), dtype=torch.int64, device=cdev)
            dist.all_to_all_single(ridx, lidx_sorted.to(cdev), self.recv_splits, self.send_splits, group=group)
            self.recv_idx = ridx.to(device)
        else:
            self.recv_splits = self.send_splits
            self.recv_idx = lidx_sorted
        self._send = torch.empty((self.nghost, 3), dtype=dtype, device=device)
        self._recv = torch.empty((self.recv_idx.numel(), 3), dtype=dtype, device=device)

    def reset_single(self, owner_idx: torch.Tensor, shift: torch.Tensor) -> None:
        """Single-rank periodic case: replace the ghost set (what LAMMPS' Comm::borders() does at re-neighbouring).
        owner_idx[nghost]: owned atom each ghost images; shift[nghost,3]: its image displacement."""
        assert self.world == 1
        self.nghost = int(owner_idx.numel())
        self.ghost_perm = torch.arange(self.nghost, device=self.device)
        self.identity_perm = True   # ghosts already in message order: the permutation gathers/scatters are skipped
        self.recv_idx = owner_idx
        self.shift = shift.to(self.dtype)
        self.send_splits = self.recv_splits = [self.nghost]
        self._send = torch.empty((self.nghost, 3), dtype=self.dtype, device=self.device)
        self._recv = torch.empty((self.nghost, 3), dtype=self.dtype, device=self.device)

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.host_staged:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def reverse_add(self, f: torch.Tensor) -> None:
        """f: [ntotal,3]; adds every ghost's force into its owner's row (on whichever rank that is)."""
        if self.nghost == 0 and self.recv_idx.numel() == 0:
            return
        if self.world == 1 and self.identity_perm:
            f[: self.nlocal].index_add_(0, self.recv_idx, f[self.nlocal:])
            return
        torch.index_select(f[self.nlocal:], 0, self.ghost_perm, out=self._send)
        if self.world > 1:
            self._a2a(self._recv, self._send, self.recv_splits, self.send_splits)
            f[: self.nlocal].index_add_(0, self.recv_idx, self._recv)
        else:
            f[: self.nlocal].index_add_(0, self.recv_idx, self._send)

    def forward_positions(self, x: torch.Tensor) -> None:
        """x: [ntotal,3]; refreshes ghost rows from their owners' rows (+ periodic image shift)."""
        if self.nghost == 0 and self.recv_idx.numel() == 0:
            return
        torch.index_select(x[: self.nlocal], 0, self.recv_idx, out=self._recv)
        if self.world == 1 and self.identity_perm:
            torch.add(self._recv, self.shift, out=x[self.nlocal:])
            return
        if self.world > 1:
            self._a2a(self._send, self._recv, self.send_splits, self.recv_splits)
            x[self.nlocal:].index_copy_(0, self.ghost_perm, self._send + self.shift)
        else:
            x[self.nlocal:].index_copy_(0, self.ghost_perm, self._recv + self.shift)


def grid_for(nranks: int):
    """LAMMPS-style processor grids used by the reference's scaling runs (examples/benchmark/submit_scaling.py:13-21)."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(nranks) or _factor3(nranks)


def _factor3(n):
    best = (n, 1, 1)
    for a in range(1, n + 1):
        if n % a:
            continue
        for b in range(1, n // a + 1):
            if (n // a) % b:
                continue
            c = n // a // b
            if max(a, b, c) < max(best):
                best = tuple(sorted((a, b, c), reverse=True))
    return best


class DomainComm:
    """LAMMPS ``Comm`` for a brick decomposition, with every array on the device: what ``Comm::exchange`` (owned atoms
    that left the sub-box migrate to their new owner), ``Comm::borders`` (ghost atoms within ``cutghost`` of the sub-box,
    periodic images included), ``Comm::forward_comm`` (ghost positions) and the pair style's ``comm->reverse_comm(this)``
    (ghost forces summed into their owners, src/pair_ani.cpp:197-201,461-484) do on the host in the reference's runs.

    The ghost set is the one LAMMPS' six face swaps produce — every atom image inside the sub-box widened by
    ``cutghost`` — but built in ONE step: each rank tests its owned atoms against the (destination brick, image shift)
    combinations that can reach them, and each peer receives one message per direction (`all_to_all_single` with per-peer
    splits: RCCL on the GPU box, gloo in the CPU tests).  Ghosts arrive grouped by sending rank, so the forward exchange
    writes the ghost block in place and the reverse exchange is one `index_add_` on the sender's list.

    Layout after ``borders``: ``x[:nlocal]`` owned atoms, ``x[nlocal:]`` ghosts in arrival order.
    """

    def __init__(self, grid, box_lo, box_len, cutghost, device, group=None, periodic=(True, True, True), native=None,
                 force_collectives=False):
        """native: an ani_hip.NativeComm (include/ani_comm.h) -- every message then travels as grouped ncclSend / ncclRecv
        inside libani_hip.so instead of torch.distributed collectives (device arrays only); force_collectives: a single rank
        takes the several-rank code paths (collectives with one participant) -- what lets one GPU exercise them."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.native = native
        if native is not None:
            assert (native.world, native.rank) == (self.world, self.rank), "the native communicator spans other ranks"
        self.multi = self.world > 1 or bool(force_collectives)
        px, py, pz = grid
        assert px * py * pz == self.world, (grid, self.world)
        self.grid = (px, py, pz)
        self.device = device
        self.cut = float(cutghost)
        self.periodic = tuple(bool(v) for v in periodic)
        self.lo_np = np.asarray(box_lo, dtype=np.float64)
        self.len_np = np.asarray(box_len, dtype=np.float64)
        self.box_lo = torch.as_tensor(self.lo_np, device=device)
        self.box_len = torch.as_tensor(self.len_np, device=device)
        self.host_staged = (native is None and self.world > 1 and dist.get_backend(group) == "gloo" and
                            torch.device(device).type == "cuda")
        r = self.rank
        self.me = (r % px, (r // px) % py, r // (px * py))   # harness/lmp_harness.cpp: rank = ix + px*(iy + py*iz)
        P = np.array(self.grid, dtype=np.float64)
        self.sub_lo = self.lo_np + self.len_np * np.array(self.me) / P
        self.sub_hi = self.lo_np + self.len_np * (np.array(self.me) + 1) / P
        # per dimension: the (brick, image shift) combinations whose widened brick can contain an image of an atom owned
        # here; [lo, hi) is the interval of the UNSHIFTED coordinate that qualifies
        combos = []
        for d in range(3):
            nimg = int(np.ceil(self.cut / self.len_np[d])) if self.periodic[d] else 0
            cd = []
            for b in range(self.grid[d]):
                blo = self.lo_np[d] + self.len_np[d] * b / self.grid[d]
                bhi = self.lo_np[d] + self.len_np[d] * (b + 1) / self.grid[d]
                for s in range(-nimg, nimg + 1):
                    lo, hi = blo - self.cut - s * self.len_np[d], bhi + self.cut - s * self.len_np[d]
                    if hi > self.sub_lo[d] and lo < self.sub_hi[d]:
                        cd.append((b, s, lo, hi))
            combos.append(cd)
        full = []
        for ix, (bx, sx, _, _) in enumerate(combos[0]):
            for iy, (by, sy, _, _) in enumerate(combos[1]):
                for iz, (bz, sz, _, _) in enumerate(combos[2]):
                    if (bx, by, bz) == self.me and (sx, sy, sz) == (0, 0, 0):
                        continue   # the owned atoms themselves
                    full.append((bx + px * (by + py * bz), ix, iy, iz, sx, sy, sz))
        full.sort(key=lambda t: t[0])   # by destination rank: the hits then come out grouped by peer
        self._dest = torch.tensor([t[0] for t in full], dtype=torch.long, device=device)
        self._cidx = [torch.tensor([t[1 + d] for t in full], dtype=torch.long, device=device) for d in range(3)]
        self._shift = torch.tensor([[t[4], t[5], t[6]] for t in full], dtype=torch.float64, device=device) * self.box_len
        # [combos, 1, 3]: interval of the unshifted position that combination accepts
        self._clo = torch.tensor([[combos[d][t[1 + d]][2] for d in range(3)] for t in full], dtype=torch.float64,
                                 device=device).reshape(-1, 1, 3)
        self._chi = torch.tensor([[combos[d][t[1 + d]][3] for d in range(3)] for t in full], dtype=torch.float64,
                                 device=device).reshape(-1, 1, 3)
        self.nlocal = 0
        self.nghost = 0
        self.send_idx = torch.zeros(0, dtype=torch.long, device=device)
        self.send_shift = torch.zeros((0, 3), dtype=torch.float64, device=device)
        self.send_splits = [0] * self.world
        self.recv_splits = [0] * self.world

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _counts(self, send_counts):
        """exchange per-peer message sizes (one host round trip; rebuild steps only)"""
        if not self.multi:
            return list(send_counts)
        if self.native is not None:
            return self.native.exchange_counts(send_counts, stream=self._stream())
        sc = torch.as_tensor(send_counts, dtype=torch.int64, device="cpu" if self.host_staged or dist.get_backend(self.group) == "gloo" else self.device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        return rc.cpu().tolist()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if torch.device(self.device).type == "cuda" else None

    def _a2a(self, inp, out_splits, in_splits, out=None):
        """out: optional contiguous destination (e.g. the ghost block of the position array: received in place)"""
        if out is None:
            out = torch.empty((int(sum(out_splits)),) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        if not self.multi:
            out.copy_(inp)
        elif self.native is not None:
            inp = inp.contiguous()
            assert out.is_contiguous()
            item = inp.element_size() * int(np.prod(inp.shape[1:], dtype=np.int64))
            self.native.alltoallv(inp.data_ptr(), in_splits, out.data_ptr(), out_splits, item, stream=self._stream())
            self._inflight = inp   # the send buffer must outlive the enqueued transfer (stream-ordered allocator: same stream)
        elif self.host_staged:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu().contiguous(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)
        return out

    # ---- Comm::exchange -------------------------------------------------------------------------------------------
    def exchange(self, x, *per_atom):
        """x: [nlocal,3] owned positions (any image); per_atom: further [nlocal,...] tensors that travel with their atom.
        Wraps positions into the box (Domain::pbc) and hands every atom to the rank whose brick now holds it.
        Returns the new (x, *per_atom) of this rank; atoms that stay keep their relative order."""
        L, lo = self.box_len, self.box_lo
        if all(self.periodic):
            x = x - torch.floor((x - lo) / L) * L
            x = torch.where(x >= lo + L, lo.expand_as(x), x)   # a coordinate that rounds up onto the upper face
        else:
            x = x.clone()
            for d in range(3):
                if self.periodic[d]:
                    x[:, d] -= torch.floor((x[:, d] - lo[d]) / L[d]) * L[d]
                    x[:, d] = torch.where(x[:, d] >= lo[d] + L[d], lo[d], x[:, d])
        if not self.multi:
            return (x,) + tuple(per_atom)
        P = torch.tensor(self.grid, dtype=torch.float64, device=x.device)
        b = torch.floor((x - lo) / L * P).long()
        b = torch.minimum(torch.clamp(b, min=0), torch.tensor(self.grid, device=x.device) - 1)
        dest = b[:, 0] + self.grid[0] * (b[:, 1] + self.grid[1] * b[:, 2])
        order = torch.argsort(dest, stable=True)
        send = torch.bincount(dest, minlength=self.world).cpu().tolist()
        recv = self._counts(send)
        out = []
        for t in (x,) + tuple(per_atom):
            out.append(self._a2a(t[order], recv, send))
        return tuple(out)

    # ---- Comm::borders --------------------------------------------------------------------------------------------
    def borders(self, x, species):
        """x: [nlocal,3] owned positions inside this rank's brick, species: [nlocal].  Selects what every peer (and this
        rank itself, for periodic self-images) needs as ghosts, exchanges it, and returns (x_all, species_all) with the
        ghosts appended."""
        n = x.shape[0]
        if self._dest.numel() == 0:
            hit = torch.zeros((0, 2), dtype=torch.long, device=x.device)
        else:
            xb = x[None, :, :]
            mask = ((xb >= self._clo) & (xb < self._chi)).all(dim=2)   # [combos, n]
            hit = mask.nonzero()                                       # combo-major, atoms ascending
        self.send_idx = hit[:, 1].contiguous()
        self.send_shift = self._shift[hit[:, 0]]
        if self.world == 1:
            send = [int(hit.shape[0])]
        else:
            send = torch.bincount(self._dest[hit[:, 0]], minlength=self.world).cpu().tolist()
        self.send_splits = send
        self.recv_splits = self._counts(send)
        self.nlocal = n
        self.nghost = int(sum(self.recv_splits))
        if self.native is not None:
            self.send_shift = self.send_shift.contiguous()
            self.native.set_epoch(self.send_splits, self.recv_splits, self.send_idx, self.send_shift)
        gx = self._a2a(x[self.send_idx] + self.send_shift, self.recv_splits, self.send_splits)
        gs = self._a2a(species[self.send_idx], self.recv_splits, self.send_splits)
        return torch.cat([x, gx]).contiguous(), torch.cat([species, gs]).contiguous()

    # ---- per step -------------------------------------------------------------------------------------------------
    def forward_positions(self, x):
        """x: [nlocal + nghost, 3]; refreshes the ghost block from the owners' current positions."""
        if self.nghost == 0 and self.send_idx.numel() == 0:
            return
        if self.native is not None and self.multi:
            self.native.forward(x.data_ptr(), self.nlocal, stream=self._stream())
            return
        msg = x[: self.nlocal][self.send_idx] + self.send_shift
        if not self.multi:
            x[self.nlocal:] = msg
        else:
            x[self.nlocal:] = self._a2a(msg, self.recv_splits, self.send_splits)

    def reverse_add(self, f):
        """f: [nlocal + nghost, 3]; adds every ghost's force into its owner's row (on whichever rank that is)."""
        if self.nghost == 0 and self.send_idx.numel() == 0:
            return
        if self.native is not None and self.multi:
            self.native.reverse(f.data_ptr(), self.nlocal, stream=self._stream())
            return
        g = f[self.nlocal:]
        if self.multi:
            g = self._a2a(g, self.send_splits, self.recv_splits)
        f[: self.nlocal].index_add_(0, self.send_idx, g)
