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
