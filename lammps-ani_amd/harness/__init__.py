"""Python face of the LAMMPS stand-in harness (``lmp_harness.cpp``) plus synthetic inputs.

Nothing here is on the product path: it fabricates what LAMMPS core would hand to ``PairANI::compute``
(``src/pair_ani.cpp:70-85`` in the reference) so that tests and ``bench.py`` can drive the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblmpharness.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lmp_harness.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = C.CDLL(_SO)
        lib.hx_build.restype = C.c_void_p
        lib.hx_build.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        lib.hx_free.argtypes = [C.c_void_p]
        for name in ("hx_nlocal", "hx_nghost"):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.c_void_p]
        for name in ("hx_x", "hx_type", "hx_tag", "hx_owner_rank", "hx_owner_lidx", "hx_shift", "hx_numneigh", "hx_jlist"):
            getattr(lib, name).restype = C.c_void_p
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.hx_neigh_build.restype = C.c_int64
        lib.hx_neigh_build.argtypes = [C.c_void_p, C.c_double, C.c_int]
        lib.hx_set_x.argtypes = [C.c_void_p, C.c_void_p]
        lib.hx_sub_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


@dataclass
class System:
    """A whole periodic (or open) system before decomposition.  ``types`` are LAMMPS types (1-based)."""
    x: np.ndarray          # [N,3] float64
    types: np.ndarray      # [N] int32, 1-based
    boxlo: np.ndarray      # [3]
    boxhi: np.ndarray      # [3]
    periodic: Tuple[bool, bool, bool] = (True, True, True)

    @property
    def natoms(self) -> int:
        return len(self.types)


@dataclass
class RankInput:
    """What one MPI rank's PairANI::compute sees (src/pair_ani.cpp:70-85) + the comm maps LAMMPS keeps."""
    nlocal: int
    nghost: int
    x: np.ndarray            # [ntotal,3] float64
    types: np.ndarray        # [ntotal] int32 (1-based)
    tag: np.ndarray          # [ntotal] global index
    owner_rank: np.ndarray   # [nghost]
    owner_lidx: np.ndarray   # [nghost]
    shift: np.ndarray        # [nghost,3]
    ilist: np.ndarray        # [nlocal] int32 (identity)
    numneigh: np.ndarray     # [nlocal] int32
    jlist: np.ndarray        # [npairs] int32, flattened in ilist order
    half: bool

    @property
    def ntotal(self) -> int:
        return self.nlocal + self.nghost

    @property
    def species(self) -> np.ndarray:
        """0-based species, ``type - 1`` as src/pair_ani.cpp:110."""
        return (self.types.astype(np.int64) - 1)

    @property
    def npairs(self) -> int:
        return int(self.jlist.shape[0])

    def atom_index12(self) -> np.ndarray:
        """Half list in the reference layout [0:n]=i, [n:2n]=j (src/pair_ani.cpp:144-145)."""
        assert self.half
        i = np.repeat(self.ilist.astype(np.int64), self.numneigh)
        return np.concatenate([i, self.jlist.astype(np.int64)])


def decompose(sys: System, grid=(1, 1, 1), rank: int = 0, cutoff: float = 5.1, skin: float = 2.0,
              half: bool = False, x_override: Optional[np.ndarray] = None) -> RankInput:
    """Atoms + ghosts + neighbour list of one rank of a px*py*pz brick decomposition."""
    lib = _load()
    x = np.ascontiguousarray(sys.x if x_override is None else x_override, dtype=np.float64)
    t = np.ascontiguousarray(sys.types, dtype=np.int32)
    lo = np.ascontiguousarray(sys.boxlo, dtype=np.float64)
    hi = np.ascontiguousarray(sys.boxhi, dtype=np.float64)
    per = np.ascontiguousarray(np.array(sys.periodic, dtype=np.int32))
    cut = cutoff + skin
    h = lib.hx_build(sys.natoms, x.ctypes.data, t.ctypes.data, lo.ctypes.data, hi.ctypes.data, per.ctypes.data,
                     grid[0], grid[1], grid[2], rank, cut)
    try:
        nl, ng = lib.hx_nlocal(h), lib.hx_nghost(h)
        nt = nl + ng
        npairs = lib.hx_neigh_build(h, cut, 1 if half else 0)
        out = RankInput(
            nlocal=nl, nghost=ng,
            x=_view(lib.hx_x(h), nt * 3, np.float64).reshape(nt, 3),
            types=_view(lib.hx_type(h), nt, np.int32),
            tag=_view(lib.hx_tag(h), nt, np.int32),
            owner_rank=_view(lib.hx_owner_rank(h), ng, np.int32),
            owner_lidx=_view(lib.hx_owner_lidx(h), ng, np.int32),
            shift=_view(lib.hx_shift(h), ng * 3, np.int32).reshape(ng, 3),
            ilist=np.arange(nl, dtype=np.int32),
            numneigh=_view(lib.hx_numneigh(h), nl, np.int32),
            jlist=_view(lib.hx_jlist(h), npairs, np.int32),
            half=half)
    finally:
        lib.hx_free(h)
    return out


# ------------------------------------------------------------------------------------------------------
# inputs
# ------------------------------------------------------------------------------------------------------

def read_lammps_data(path: str) -> System:
    """Minimal reader for ``atom_style atomic`` data files such as tests/golden/water-0.8nm.data."""
    lo, hi = np.zeros(3), np.zeros(3)
    rows = []
    natoms = None
    section = None
    with open(path) as f:
        for line in f:
            s = line.split("#")[0].strip()
            if not s:
                continue
            tok = s.split()
            if len(tok) >= 2 and tok[1] == "atoms":
                natoms = int(tok[0])
            elif len(tok) == 4 and tok[2] in ("xlo", "ylo", "zlo"):
                k = "xyz".index(tok[2][0])
                lo[k], hi[k] = float(tok[0]), float(tok[1])
            elif tok[0] in ("Masses", "Atoms", "Velocities"):
                section = tok[0]
            elif section == "Atoms" and len(tok) >= 5:
                rows.append((int(tok[0]), int(tok[1]), float(tok[2]), float(tok[3]), float(tok[4])))
    rows.sort()
    assert natoms == len(rows), (natoms, len(rows))
    arr = np.array(rows)
    return System(arr[:, 2:5].astype(np.float64).copy(), arr[:, 1].astype(np.int32), lo, hi)


def water_box(n_atoms: int, seed: int = 12345, density: float = 0.98, type_H: int = 1, type_O: int = 4,
              min_oo: float = 2.4, min_contact: float = 1.4) -> System:
    """Synthetic liquid-water-like box: n_atoms/3 rigid waters (r_OH 0.9572 A, 104.52 deg),
    centres on a jittered simple-cubic lattice (sigma 0.3 A, O-O >= min_oo), random orientations re-drawn until no
    intermolecular pair is closer than min_contact; density as in
    examples/benchmark/data/water/prepare/generate_pdb.py:29-48 (0.98 g/cm^3).  Atom order O,H,H per molecule.
    Without the two rejections a few pairs per 10^4 atoms sit at 0.4-0.6 A and carry forces of 10^3 kcal/mol/A,
    which no equilibrated water box has."""
    from scipy.spatial import cKDTree
    assert n_atoms % 3 == 0
    nmol = n_atoms // 3
    mass_g = nmol * (15.999 + 2 * 1.008) / 6.0221408e23
    L = (mass_g / density * 1e24) ** (1.0 / 3.0)
    rng = np.random.default_rng(seed)
    ncell = int(np.ceil(nmol ** (1.0 / 3.0)))
    a = L / ncell
    idx = rng.permutation(ncell ** 3)[:nmol]
    site = (np.stack([idx % ncell, (idx // ncell) % ncell, idx // (ncell * ncell)], 1) + 0.5) * a
    centres = site + rng.normal(0.0, 0.3, size=(nmol, 3))
    for _ in range(500):
        w = np.mod(centres, L)
        w[w >= L] = 0.0
        pairs = cKDTree(w, boxsize=L).query_pairs(min_oo, output_type="ndarray")
        if len(pairs) == 0:
            break
        redo = np.unique(pairs[:, 1])
        centres[redo] = site[redo] + rng.normal(0.0, 0.3, size=(len(redo), 3))

    roh, ang = 0.9572, np.deg2rad(104.52)
    h1 = np.array([roh * np.sin(ang / 2), roh * np.cos(ang / 2), 0.0])
    h2 = np.array([-roh * np.sin(ang / 2), roh * np.cos(ang / 2), 0.0])

    def random_rotations(k):
        q = rng.normal(size=(k, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        w_, x_, y_, z_ = q.T
        return np.stack([
            np.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)], 1),
            np.stack([2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_)], 1),
            np.stack([2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)], 1)], 1)

    def place(Rm, c):
        out = np.empty((len(c), 3, 3))
        out[:, 0] = c
        out[:, 1] = c + Rm @ h1
        out[:, 2] = c + Rm @ h2
        return out

    pos = place(random_rotations(nmol), centres)
    for _ in range(500):
        flat = np.mod(pos.reshape(-1, 3), L)
        flat[flat >= L] = 0.0
        pairs = cKDTree(flat, boxsize=L).query_pairs(min_contact, output_type="ndarray")
        bad = pairs[(pairs[:, 0] // 3) != (pairs[:, 1] // 3)]
        if len(bad) == 0:
            break
        redo = np.unique(bad[:, 1] // 3)
        pos[redo] = place(random_rotations(len(redo)), centres[redo])
    lo, hi = np.full(3, -L / 2), np.full(3, L / 2)
    x = lo + np.mod(pos.reshape(-1, 3), L)
    x[x >= hi] = lo[0]
    types = np.tile(np.array([type_O, type_H, type_H], dtype=np.int32), nmol)
    return System(x, types, lo, hi)


def combustion_box(n_atoms: int, seed: int = 12345, density: float = 0.25, type_H: int = 1, type_C: int = 2,
                   type_O: int = 4) -> System:
    """Synthetic CH4 : O2 = 1 : 2 gas at 0.25 g/cm^3 (the reactive box of
    examples/combustion/prepare_system/generate_pdb.py:27-52): rigid methane (r_CH 1.09 A, tetrahedral) and oxygen
    (r_OO 1.21 A) molecules, centres on a jittered simple-cubic lattice (sigma 0.5 A), random orientations.  n_atoms is
    rounded down to whole groups of 9 atoms (C H4 + 2 O2); ~26 neighbours within 7.1 A."""
    rng = np.random.default_rng(seed)
    ngroup = n_atoms // 9
    nmol = 3 * ngroup
    mass_g = ngroup * (12.011 + 4 * 1.008 + 4 * 15.999) / 6.0221408e23
    L = (mass_g / density * 1e24) ** (1.0 / 3.0)
    ncell = int(np.ceil(nmol ** (1.0 / 3.0)))
    idx = rng.permutation(ncell ** 3)[:nmol]
    centres = (np.stack([idx % ncell, (idx // ncell) % ncell, idx // (ncell * ncell)], 1) + 0.5) * (L / ncell)
    centres += rng.normal(0.0, 0.5, size=(nmol, 3))
    q = rng.normal(size=(nmol, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w_, x_, y_, z_ = q.T
    R = np.stack([np.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)], 1),
                  np.stack([2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_)], 1),
                  np.stack([2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)], 1)], 1)
    t = 1.09 / np.sqrt(3.0)
    ch4 = np.array([[0, 0, 0], [t, t, t], [t, -t, -t], [-t, t, -t], [-t, -t, t]], dtype=np.float64)
    o2 = np.array([[0.605, 0, 0], [-0.605, 0, 0]], dtype=np.float64)
    pos, types = [], []
    for g in range(ngroup):
        for k, (tmpl, ty) in enumerate(((ch4, [type_C] + [type_H] * 4), (o2, [type_O] * 2), (o2, [type_O] * 2))):
            mi = 3 * g + k
            pos.append(centres[mi] + tmpl @ R[mi].T)
            types.extend(ty)
    x = np.mod(np.concatenate(pos), L)
    x[x >= L] = 0.0
    lo, hi = np.full(3, -L / 2), np.full(3, L / 2)
    return System(x + lo, np.asarray(types, dtype=np.int32), lo, hi)


def random_box(n_atoms: int, ntypes: int, L: float, seed: int = 7, min_dist: float = 0.9) -> System:
    """Mixed-species random box (rejection on a minimum distance) — exercises every species bucket."""
    rng = np.random.default_rng(seed)
    pts = []
    cell = {}
    inv = 1.0 / min_dist
    ncell = max(1, int(L * inv))

    def key(p):
        return tuple((np.floor(p * ncell / L).astype(int)) % ncell)

    tries = 0
    while len(pts) < n_atoms and tries < 200 * n_atoms:
        tries += 1
        p = rng.uniform(0, L, 3)
        k = key(p)
        ok = True
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    kk = ((k[0] + dx) % ncell, (k[1] + dy) % ncell, (k[2] + dz) % ncell)
                    for q in cell.get(kk, ()):
                        dd = p - q
                        dd -= L * np.round(dd / L)
                        if dd @ dd < min_dist * min_dist:
                            ok = False
        if ok:
            pts.append(p)
            cell.setdefault(k, []).append(p)
    assert len(pts) == n_atoms, "box too dense for min_dist"
    x = np.array(pts) - L / 2
    types = rng.integers(1, ntypes + 1, size=n_atoms).astype(np.int32)
    return System(x, types, np.full(3, -L / 2), np.full(3, L / 2))


def spatial_sort(sys: System, binsize: float = 3.55) -> System:
    """Reorder atoms by spatial bins, as LAMMPS' default ``atom_modify sort 1000 <half the neighbour cutoff>`` does
    for every run of the reference (bins of (5.1 + 2.0) / 2 A), so that neighbours are close in memory."""
    L = sys.boxhi - sys.boxlo
    nb = np.maximum(1, np.floor(L / binsize).astype(np.int64))
    b = np.minimum(((sys.x - sys.boxlo) / L * nb).astype(np.int64), nb - 1)
    key = (b[:, 2] * nb[1] + b[:, 1]) * nb[0] + b[:, 0]
    order = np.argsort(key, kind="stable")
    return System(sys.x[order].copy(), sys.types[order].copy(), sys.boxlo, sys.boxhi, sys.periodic)
