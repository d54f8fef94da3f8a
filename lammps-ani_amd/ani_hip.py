"""ctypes binding of the C ABI in include/ani_hip.h (libani_hip.so).

Mirrors the reference's ``class ANI`` (src/ani_csrc/ani.h:11-85): construct with the pair_style arguments, call
``compute`` each step with ``ago`` telling whether the neighbour list was rebuilt.  There is no CPU fallback:
if the library is missing or no GPU is visible this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# ANI_HIP_LIB selects an alternative build of the same library (kernel A/B experiments); never a different backend
LIB_PATH = os.environ.get("ANI_HIP_LIB") or os.path.join(_CSRC, "libani_hip.so")
HARTREE2KCALMOL = 627.5094738898777

EXPORTS = ["ani_create", "ani_destroy", "ani_last_error", "ani_num_models", "ani_use_num_models", "ani_num_species",
           "ani_aev_length", "ani_cutoff_radial", "ani_cutoff_angular", "ani_compute_full", "ani_compute_half",
           "ani_compute_full_device", "ani_build_list_device", "ani_build_list", "ani_debug_list", "ani_debug_get", "ani_debug_read", "ani_debug_colmap", "ani_set_option", "ani_phase_timing", "ani_phase_times",
           "ani_trace_push", "ani_trace_pop", "ani_trace_mark", "ani_step_begin", "ani_step_ghosts_ready", "ani_step_finish",
           "ani_debug_fused_stamps", "ani_attach_comm", "ani_debug_fused_schedule", "ani_debug_fused_schedule_halves", "ani_last_mlp_kernel", "ani_host_register", "ani_host_unregister", "ani_set_ghost_fold", "ani_stage_ghost_fold"]
# include/ani_comm.h: the device-side ghost exchange over RCCL
COMM_EXPORTS = ["ani_comm_get_unique_id", "ani_comm_create", "ani_comm_create_local", "ani_comm_destroy", "ani_comm_last_error", "ani_comm_rank",
                "ani_comm_size", "ani_comm_plan", "ani_comm_exchange_counts", "ani_comm_alltoallv", "ani_comm_set_epoch",
                "ani_comm_set_epoch_host", "ani_comm_set_ghost_order", "ani_comm_forward", "ani_comm_reverse", "ani_comm_reverse_send", "ani_comm_reverse_unpack",
                "ani_comm_allreduce_f64", "ani_comm_set_option", "ani_comm_get_stat"]


class AniError(RuntimeError):
    pass


class DebugView(C.Structure):
    _fields_ = [("nlocal", C.c_int), ("ntotal", C.c_int), ("nrows", C.c_int), ("npairs", C.c_int64),
                ("d_aev", C.c_void_p), ("d_gaev", C.c_void_p), ("d_row_of_centre", C.c_void_p),
                ("species_count", C.c_int * 16), ("aev_stride", C.c_int), ("aev_active_length", C.c_int), ("error_flags", C.c_int)]


def build(force: bool = False) -> str:
    """Compile libani_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  force: rebuild every object
    (make -B) whatever the timestamps say.  Returns the library path; build.last_commands holds the compiler command
    lines make ran (empty when everything was up to date)."""
    args = ["make", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    out = subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    build.last_commands = [ln for ln in out.splitlines() if ln.lstrip().startswith(("hipcc", "g++", "gcc"))]
    return LIB_PATH


build.last_commands = []


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AniError(f"{LIB_PATH} is missing: build it (python -c 'import __graft_entry__ as g; g.build()'); "
                           "there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.ani_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ani_destroy.argtypes = [C.c_void_p]
        L.ani_last_error.restype = C.c_char_p
        L.ani_last_error.argtypes = [C.c_void_p]
        for n in ("ani_num_models", "ani_use_num_models", "ani_num_species", "ani_aev_length"):
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("ani_cutoff_radial", "ani_cutoff_angular"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = C.c_double
        L.ani_compute_full.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.ani_compute_half.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.ani_compute_full_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ani_step_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.ani_step_ghosts_ready.argtypes = [C.c_void_p, C.c_void_p]
        L.ani_step_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.ani_build_list_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
        L.ani_build_list.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_int64)]
        L.ani_debug_list.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3
        L.ani_debug_get.argtypes = [C.c_void_p, C.POINTER(DebugView)]
        L.ani_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.ani_debug_colmap.argtypes = [C.c_void_p, C.c_void_p]
        L.ani_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ani_last_mlp_kernel.argtypes = [C.c_void_p]
        L.ani_set_ghost_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_stage_ghost_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ani_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.ani_host_unregister.argtypes = [C.c_void_p]
        L.ani_last_mlp_kernel.restype = C.c_char_p
        # include/ani_md.h: kernels of the LAMMPS-free timestep loop (not part of the drop-in boundary)
        L.ani_md_initial_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
        L.ani_md_final_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.ani_md_final_initial_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]
        L.ani_md_wrap_positions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_md_ghost_shell_count.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.ani_md_ghost_shell_fill.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        L.ani_md_append_ghosts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_md_check.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ani_md_forward_ghosts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ani_md_reverse_ghosts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ani_md_reverse_ghosts_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ani_md_pack_ghosts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ani_md_unpack_reverse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        # include/ani_comm.h
        L.ani_comm_get_unique_id.argtypes = [C.c_void_p]
        L.ani_comm_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ani_comm_destroy.argtypes = [C.c_void_p]
        L.ani_comm_last_error.restype = C.c_char_p
        L.ani_comm_last_error.argtypes = [C.c_void_p]
        L.ani_comm_rank.argtypes = [C.c_void_p]
        L.ani_comm_size.argtypes = [C.c_void_p]
        L.ani_comm_plan.argtypes = [C.c_int] + [C.c_void_p] * 6
        L.ani_comm_exchange_counts.argtypes = [C.c_void_p] * 4
        L.ani_comm_alltoallv.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.ani_comm_set_epoch.argtypes = [C.c_void_p] * 5
        L.ani_comm_set_ghost_order.argtypes = [C.c_void_p, C.c_void_p]
        L.ani_comm_set_epoch_host.argtypes = [C.c_void_p] * 6
        L.ani_attach_comm.argtypes = [C.c_void_p, C.c_void_p]
        L.ani_md_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ani_md_scatter_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ani_comm_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_comm_reverse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_comm_reverse_send.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ani_comm_reverse_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ani_comm_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ani_comm_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ani_comm_get_stat.argtypes = [C.c_void_p, C.c_char_p]
        L.ani_comm_get_stat.restype = C.c_longlong
        L.ani_phase_timing.argtypes = [C.c_void_p, C.c_int]
        L.ani_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


class ANI:
    """``ANI(model_file, local_rank, use_num_models, use_cuaev, use_fullnbr, use_single)`` — src/ani_csrc/ani.h:31-36."""

    def __init__(self, model_file: str, local_rank: int = 0, use_num_models: int = -1, use_cuaev: bool = True,
                 use_fullnbr: bool = True, use_single: bool = True):
        self._lib = lib()
        self._h = C.c_void_p()
        rc = self._lib.ani_create(model_file.encode(), local_rank, use_num_models, int(use_cuaev), int(use_fullnbr),
                                  int(use_single), C.byref(self._h))
        if rc != 0:
            raise AniError(f"ani_create failed ({rc}): {self._lib.ani_last_error(None).decode()}")
        self.use_cuaev, self.use_fullnbr, self.use_single = use_cuaev, use_fullnbr, use_single
        self.num_models = self._lib.ani_num_models(self._h)
        self.use_num_models = self._lib.ani_use_num_models(self._h)
        self.aev_length = self._lib.ani_aev_length(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ani_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AniError(f"libani_hip error {rc}: {self._lib.ani_last_error(self._h).decode()}")

    def compute(self, inp, ago: int = 0, eflag_atom: bool = True, vflag: bool = True, force_into=None):
        """Host-pointer entry points with a harness.RankInput (full or half list).  Returns a dict like the oracle's.
        force_into: a C-contiguous float64 [ntotal, 3] array handed over as out_force (option out_force_accumulate adds into it)."""
        nt, nl = inp.ntotal, inp.nlocal
        species = np.ascontiguousarray(inp.species, dtype=np.int64)
        x = np.ascontiguousarray(inp.x, dtype=np.float64)
        e = np.zeros(1)
        f = np.full((nt, 3), np.nan) if force_into is None else force_into
        assert f.dtype == np.float64 and f.shape == (nt, 3) and f.flags.c_contiguous
        ea = np.zeros(nl)
        vir = np.zeros(9)
        if inp.half:
            a12 = np.ascontiguousarray(inp.atom_index12(), dtype=np.int64)
            rc = self._lib.ani_compute_half(self._h, nt, nl, species.ctypes.data, x.ctypes.data, inp.npairs,
                                            a12.ctypes.data, ago, int(eflag_atom), int(vflag), e.ctypes.data,
                                            f.ctypes.data, ea.ctypes.data, vir.ctypes.data)
        else:
            il = np.ascontiguousarray(inp.ilist, dtype=np.int32)
            nn = np.ascontiguousarray(inp.numneigh, dtype=np.int32)
            jl = np.ascontiguousarray(inp.jlist, dtype=np.int32)
            rc = self._lib.ani_compute_full(self._h, nt, nl, species.ctypes.data, x.ctypes.data, inp.npairs,
                                            il.ctypes.data, jl.ctypes.data, nn.ctypes.data, ago, int(eflag_atom),
                                            int(vflag), e.ctypes.data, f.ctypes.data, ea.ctypes.data, vir.ctypes.data)
        self._check(rc)
        return dict(energy=float(e[0]), force=f, eatom=ea, virial=vir.reshape(3, 3))

    def compute_device(self, ntotal, nlocal, d_species, d_x, npairs, d_ilist, d_jlist, d_numneigh, ago, d_f, d_ev,
                       d_eatom=None, eflag_atom=False, vflag=False, stream=None):
        """Device-resident step; arguments are raw device addresses (e.g. torch tensor .data_ptr())."""
        rc = self._lib.ani_compute_full_device(self._h, ntotal, nlocal, d_species, d_x, npairs, d_ilist, d_jlist,
                                               d_numneigh, ago, int(eflag_atom), int(vflag), d_f, d_ev, d_eatom, stream)
        self._check(rc)

    def step_begin(self, ntotal, nlocal, d_x, d_f, d_ev, d_eatom=None, eflag_atom=False, vflag=False, stream=None):
        """Split device-resident step, part 1 of 3 (include/ani_hip.h): needs the owned atoms' positions only."""
        self._check(self._lib.ani_step_begin(self._h, ntotal, nlocal, d_x, int(eflag_atom), int(vflag), d_f, d_ev, d_eatom, stream))

    def step_ghosts_ready(self, stream=None):
        """Part 2: the ghost positions are in place; afterwards the ghost rows of the force array are final."""
        self._check(self._lib.ani_step_ghosts_ready(self._h, stream))

    def step_finish(self, stream=None):
        """Part 3: forces of the owned atoms, energy, virial."""
        self._check(self._lib.ani_step_finish(self._h, stream))

    def build_list_device(self, ntotal, nlocal, d_species, d_x, cutneigh, lo, hi, stream=None) -> int:
        """Device-side full neighbour list (ani_build_list_device); returns the pair count.  Follow with
        ``compute_device(..., ago=1)`` and null list pointers."""
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        n = C.c_int64()
        self._check(self._lib.ani_build_list_device(self._h, ntotal, nlocal, d_species, d_x, float(cutneigh),
                                                    lo.ctypes.data, hi.ctypes.data, C.byref(n), stream))
        return n.value

    def build_list(self, species, x, nlocal: int, cutneigh: float, lo=None, hi=None) -> int:
        """Host-array form (ani_build_list): species[ntotal], x[ntotal,3] with owned atoms first; returns the pair
        count.  Follow with the host-pointer compute at ago != 0."""
        species = np.ascontiguousarray(species, dtype=np.int64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if lo is None:   # per column: numpy's axis-0 reduction of an [n,3] array is ~10x slower
            lo = [x[:, k].min() - 0.25 for k in range(3)]
        if hi is None:
            hi = [x[:, k].max() + 0.25 for k in range(3)]
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        n = C.c_int64()
        self._check(self._lib.ani_build_list(self._h, x.shape[0], nlocal, species.ctypes.data, x.ctypes.data, float(cutneigh),
                                             lo.ctypes.data, hi.ctypes.data, C.byref(n)))
        return n.value

    def debug_list(self, nlocal: int):
        """(numneigh[nlocal], jlist[npairs]) of the list installed in the handle, as host arrays."""
        pn, po, pj = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._lib.ani_debug_list(self._h, C.byref(pn), C.byref(po), C.byref(pj)))
        nn = self.debug_read(pn, (nlocal,), np.int32)
        if nlocal == 0:
            return nn, np.zeros(0, np.int32)
        # dense segments, or rows of a fixed capacity (the one-kernel build): gathered into dense segments here
        off = self.debug_read(po, (nlocal + 1,), np.int32)
        span = self.debug_read(pj, (int((off[:-1] + nn).max()),), np.int32)
        if np.array_equal(off[1:] - off[:-1], nn):
            return nn, span[: int(nn.sum())]
        idx = np.repeat(off[:-1] - np.concatenate([[0], np.cumsum(nn)[:-1]]), nn) + np.arange(int(nn.sum()))
        return nn, span[idx]

    def debug_view(self) -> DebugView:
        v = DebugView()
        self._check(self._lib.ani_debug_get(self._h, C.byref(v)))
        return v

    def debug_read(self, d_ptr, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.ani_debug_read(self._h, d_ptr, out.ctypes.data, out.nbytes))
        return out

    def colmap(self) -> np.ndarray:
        v = self.debug_view()
        out = np.zeros(v.aev_active_length, dtype=np.int32)
        self._check(self._lib.ani_debug_colmap(self._h, out.ctypes.data))
        return out

    def set_option(self, name: str, value: int):
        self._check(self._lib.ani_set_option(self._h, name.encode(), int(value)))

    def set_ghost_fold(self, d_owner, d_shift, nghost: int, stream=None):
        """ani_set_ghost_fold: device addresses of owner[nghost] (int64) and shift[nghost][3] (float64); None clears"""
        self._check(self._lib.ani_set_ghost_fold(self._h, d_owner, d_shift, int(nghost), stream))

    def stage_ghost_fold(self, d_owner, d_shift, nghost: int):
        """ani_stage_ghost_fold: the fold of the list the NEXT build_list* call builds (checked behind that build's synchronisation)"""
        self._check(self._lib.ani_stage_ghost_fold(self._h, d_owner, d_shift, int(nghost)))

    def last_mlp_kernel(self) -> str:
        return self._lib.ani_last_mlp_kernel(self._h).decode()

    def cutoffs(self):
        """(Rcr, Rca) of the model"""
        return float(self._lib.ani_cutoff_radial(self._h)), float(self._lib.ani_cutoff_angular(self._h))

    def attach_comm(self, comm):
        """ani_attach_comm: a NativeComm (or None) whose reverse exchange the host-pointer entry points run on the device"""
        self._comm_keep = comm
        self._check(self._lib.ani_attach_comm(self._h, comm._h if comm is not None else None))

    def phase_timing(self, enable):
        """1/True: fresh accumulation; 0/False: stop recording; 2: resume without clearing."""
        self._check(self._lib.ani_phase_timing(self._h, int(enable)))

    def phase_times(self):
        ms = (C.c_double * 5)()
        n = C.c_int()
        self._check(self._lib.ani_phase_times(self._h, ms, C.byref(n)))
        return dict(aev_fwd=ms[0], mlp=ms[1], aev_bwd=ms[2], other=ms[3], compact=ms[4], calls=n.value)


def comm_plan(send_counts, recv_counts):
    """ani_comm_plan: (send_off, recv_off, nsend, nrecv) of an all-to-all with these per-peer counts.  Host arithmetic
    only: works without a GPU."""
    sc = np.ascontiguousarray(send_counts, dtype=np.int64)
    rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
    so, ro = np.zeros_like(sc), np.zeros_like(rc)
    ns, nr = C.c_int64(), C.c_int64()
    if lib().ani_comm_plan(len(sc), sc.ctypes.data, rc.ctypes.data, so.ctypes.data, ro.ctypes.data, C.addressof(ns), C.addressof(nr)) != 0:
        raise AniError("ani_comm_plan: bad counts")
    return so, ro, ns.value, nr.value


class NativeComm:
    """include/ani_comm.h: the ghost exchange between ranks as grouped ncclSend / ncclRecv inside libani_hip.so (no torch on
    the data path).  ``id_bytes``: the 128 bytes rank 0 got from ``NativeComm.unique_id()``, handed to every rank by the
    caller (``from_torch`` does it with torch.distributed, the python loop's bootstrap)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        if lib().ani_comm_get_unique_id(buf) != 0:
            raise AniError("ani_comm_get_unique_id: " + lib().ani_comm_last_error(None).decode())
        return bytes(buf)

    @classmethod
    def from_torch(cls, device_index: int, group=None):
        """bootstrap over an initialised torch.distributed group (any backend): rank 0's id is broadcast as an object.
        Every rank runs the SAME sequence of collectives whatever fails where: rank 0 broadcasts (status, id) even when it
        could not make an id, and the ranks agree on the outcome of ani_comm_create before anybody uses the communicator --
        so a caller's fallback to another transport is taken by all ranks or by none."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        box = [None]
        if rank == 0:
            try:
                box = [("ok", cls.unique_id())]
            except Exception as e:   # librccl missing, ncclGetUniqueId failed: the peers must hear about it
                box = [("error", str(e))]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        status, payload = box[0]
        if status != "ok":
            raise AniError(f"rank 0 could not make an RCCL id: {payload}")
        comm, err = None, None
        try:
            comm = cls(world, rank, payload, device_index)
        except AniError as e:
            err = e
        if world > 1:
            oks = [None] * world
            dist.all_gather_object(oks, err is None, group=group)
            if not all(oks):
                if comm is not None:
                    comm.close()
                raise AniError(f"ani_comm_create failed on rank(s) {[r for r, ok in enumerate(oks) if not ok]}"
                               + (f": {err}" if err is not None else ""))
        elif err is not None:
            raise err
        return comm

    def __init__(self, nranks: int, rank: int, id_bytes: bytes, device_index: int = 0):
        self._lib = lib()
        self._h = C.c_void_p()
        buf = C.create_string_buffer(id_bytes, 128)
        rc = self._lib.ani_comm_create(nranks, rank, buf, device_index, C.byref(self._h))
        if rc != 0:
            raise AniError(f"ani_comm_create failed ({rc}): {self._lib.ani_comm_last_error(None).decode()}")
        self.world, self.rank = nranks, rank
        self._keep = ()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ani_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AniError(f"ani_comm error {rc}: {self._lib.ani_comm_last_error(self._h).decode()}")

    def set_option(self, name: str, value: int):
        self._check(self._lib.ani_comm_set_option(self._h, name.encode(), int(value)))

    def stat(self, name: str) -> int:
        """ani_comm_get_stat: 'forward_exchanges', 'reverse_exchanges', 'alltoalls', 'broken' """
        return int(self._lib.ani_comm_get_stat(self._h, name.encode()))

    def exchange_counts(self, send_counts, stream=None):
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.zeros_like(sc)
        self._check(self._lib.ani_comm_exchange_counts(self._h, sc.ctypes.data, rc.ctypes.data, stream))
        return rc.tolist()

    def alltoallv(self, d_send, send_counts, d_recv, recv_counts, item_bytes: int, stream=None):
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        self._check(self._lib.ani_comm_alltoallv(self._h, d_send, sc.ctypes.data, d_recv, rc.ctypes.data, int(item_bytes), stream))

    def set_epoch(self, send_counts, recv_counts, send_idx, send_shift):
        """send_idx (int64) / send_shift (float64 [n,3]): device tensors; kept alive here until the next epoch"""
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        self._keep = (send_idx, send_shift)
        self._check(self._lib.ani_comm_set_epoch(self._h, sc.ctypes.data, rc.ctypes.data, send_idx.data_ptr(), send_shift.data_ptr()))

    def set_ghost_order(self, ghost_of):
        """ghost_of (int64 device tensor, or None): entry k of the rank-grouped message order is ghost ghost_of[k]"""
        self._keep = self._keep + (ghost_of,)
        self._check(self._lib.ani_comm_set_ghost_order(self._h, ghost_of.data_ptr() if ghost_of is not None else None))

    def forward(self, d_x, nlocal: int, stream=None):
        self._check(self._lib.ani_comm_forward(self._h, d_x, nlocal, stream))

    def reverse(self, d_f, nlocal: int, stream=None):
        self._check(self._lib.ani_comm_reverse(self._h, d_f, nlocal, stream))

    def reverse_send(self, d_f, nlocal: int, stream=None):
        self._check(self._lib.ani_comm_reverse_send(self._h, d_f, nlocal, stream))

    def reverse_unpack(self, d_f, stream=None):
        self._check(self._lib.ani_comm_reverse_unpack(self._h, d_f, stream))

    def allreduce(self, d_buf, n: int, op: str = "sum", stream=None):
        self._check(self._lib.ani_comm_allreduce_f64(self._h, d_buf, n, 0 if op == "sum" else 1, stream))
