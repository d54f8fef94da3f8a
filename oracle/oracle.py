"""ctypes face of the CPU oracle (oracle/ani_oracle.c).  TEST INFRASTRUCTURE ONLY — see the C file's header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    src = os.path.join(_HERE, "ani_oracle.c")
    for so in ("libani_oracle64.so", "libani_oracle32.so"):
        p = os.path.join(_HERE, so)
        if force or not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", so])


class Oracle:
    """One loaded model in the fp64 (default) or fp32 build of the oracle."""

    def __init__(self, model_file: str, use_num_models: int = -1, fp32: bool = False):
        so = os.path.join(_HERE, "libani_oracle32.so" if fp32 else "libani_oracle64.so")
        if not os.path.exists(so):
            build()
        self.lib = lib = C.CDLL(so)
        self.real = np.float32 if fp32 else np.float64
        lib.ani_oracle_load.restype = C.c_void_p
        lib.ani_oracle_load.argtypes = [C.c_char_p, C.c_int]
        lib.ani_oracle_free.argtypes = [C.c_void_p]
        lib.ani_oracle_aev_len.argtypes = [C.c_void_p]
        lib.ani_oracle_num_models.argtypes = [C.c_void_p]
        lib.ani_oracle_compute_full.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6
        lib.ani_oracle_compute_half.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_int] + [C.c_void_p] * 6
        self.h = lib.ani_oracle_load(model_file.encode(), use_num_models)
        if not self.h:
            raise RuntimeError(f"oracle: cannot load model {model_file!r} (use_num_models={use_num_models})")
        self.aev_len = lib.ani_oracle_aev_len(self.h)
        self.num_models = lib.ani_oracle_num_models(self.h)
        self.threads = lib.ani_oracle_num_threads()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ani_oracle_free(self.h)
            self.h = None

    def compute(self, inp, radial_compat: bool = False, want_aev: bool = False):
        """inp: harness.RankInput.  Returns dict(energy, force[ntotal,3], eatom[nlocal], virial[3,3], aev, gaev)."""
        nt, nl = inp.ntotal, inp.nlocal
        species = np.ascontiguousarray(inp.species, dtype=np.int64)
        x = np.ascontiguousarray(inp.x, dtype=np.float64)
        e = np.zeros(1)
        f = np.zeros((nt, 3))
        ea = np.zeros(nl)
        vir = np.zeros(9)
        aev = np.zeros((nl, self.aev_len), dtype=self.real) if want_aev else None
        gaev = np.zeros((nl, self.aev_len), dtype=self.real) if want_aev else None
        pa = aev.ctypes.data if want_aev else None
        pg = gaev.ctypes.data if want_aev else None
        if inp.half:
            a12 = np.ascontiguousarray(inp.atom_index12(), dtype=np.int64)
            rc = self.lib.ani_oracle_compute_half(self.h, nt, nl, species.ctypes.data, x.ctypes.data, inp.npairs,
                                                  a12.ctypes.data, int(radial_compat), e.ctypes.data, f.ctypes.data,
                                                  ea.ctypes.data, vir.ctypes.data, pa, pg)
        else:
            il = np.ascontiguousarray(inp.ilist, dtype=np.int32)
            nn = np.ascontiguousarray(inp.numneigh, dtype=np.int32)
            jl = np.ascontiguousarray(inp.jlist, dtype=np.int32)
            rc = self.lib.ani_oracle_compute_full(self.h, nt, nl, species.ctypes.data, x.ctypes.data, il.ctypes.data,
                                                  nn.ctypes.data, jl.ctypes.data, int(radial_compat), e.ctypes.data,
                                                  f.ctypes.data, ea.ctypes.data, vir.ctypes.data, pa, pg)
        if rc != 0:
            raise RuntimeError(f"oracle compute failed rc={rc}")
        return dict(energy=float(e[0]), force=f, eatom=ea, virial=vir.reshape(3, 3), aev=aev, gaev=gaev)
