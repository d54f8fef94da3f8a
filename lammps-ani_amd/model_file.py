"""Flat binary ANI model file (``*.anim``) — writer, reader and seeded synthetic generators.

The reference ships its potential as a TorchScript archive produced by
``models/ani_models.py:112-122`` (``torch.jit.script(LammpsANI(model)).save``) and loaded with
``torch::jit::load`` (``src/ani_csrc/ani.cpp:46``).  This build has no libtorch on the compute path, so the
same information (AEV constants, per-species network shapes, ensemble weights, self energies) is kept in a
flat little-endian file that the HIP library (``csrc/ani_model.cpp``) and the CPU oracle
(``oracle/ani_oracle.c``) both parse.

Layout (all little-endian, no padding except where stated)::

    char[8]  magic  = "ANIHIP01"
    u32      num_species S
    u32      num_models  M
    u32      num_layers  L          (affine layers per network, 4 for ANI-1x/2x)
    u32      nShfR, nShfA, nShfZ
    f64      Rcr, Rca, EtaR, EtaA, Zeta, celu_alpha
    f64[nShfR] ShfR ; f64[nShfA] ShfA ; f64[nShfZ] ShfZ
    per species s:  char[4] symbol (NUL padded) ; f64 self_energy (Hartree) ; u32[L+1] dims (dims[0] = AEV width)
    for m in models: for s in species: for l in layers:
        f32[dims[l+1]][dims[l]] W      (torch.nn.Linear layout: [out][in])
        f32[dims[l+1]]          b
    optional trailing block (pairwise repulsion):
        char[8] "REPULXTB" ; f64 cutoff (Angstrom) ; f64[S][S] y_ab ; f64[S][S] sqrt_alpha_ab ; f64[S][S] k_rep_ab

Species order is the LAMMPS type order (type-1 = species index; ``src/pair_ani.cpp:110``).
Weights of real ANI-2x are not available offline; :func:`synthetic_model` makes
shape-identical seeded stand-ins.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

MAGIC = b"ANIHIP01"


@dataclass
class AniModel:
    species: List[str]
    Rcr: float
    Rca: float
    EtaR: float
    EtaA: float
    Zeta: float
    ShfR: np.ndarray
    ShfA: np.ndarray
    ShfZ: np.ndarray
    self_energies: np.ndarray  # [S] Hartree
    dims: List[List[int]]  # [S][L+1]
    # weights[m][s][l] -> (W [out,in] f32, b [out] f32)
    weights: list = field(default_factory=list)
    celu_alpha: float = 0.1
    # optional pairwise repulsion (torchani RepulsionXTB as the reference attaches it, models/ani_models.py:50-53):
    # dict(cutoff=float, y_ab=[S,S], sqrt_alpha_ab=[S,S], k_rep_ab=[S,S]); atomic units inside (Bohr, Hartree)
    repulsion: dict = None

    @property
    def num_species(self) -> int:
        return len(self.species)

    @property
    def num_models(self) -> int:
        return len(self.weights)

    @property
    def num_layers(self) -> int:
        return len(self.dims[0]) - 1

    @property
    def radial_len(self) -> int:
        return self.num_species * len(self.ShfR)

    @property
    def angular_len(self) -> int:
        S = self.num_species
        return S * (S + 1) // 2 * len(self.ShfA) * len(self.ShfZ)

    @property
    def aev_len(self) -> int:
        return self.radial_len + self.angular_len

    def select_models(self, n: int) -> "AniModel":
        """First-n ensemble members, as ``LammpsANI.select_models`` (models/lammps_ani.py:332-343)."""
        if n is None or n < 0:
            n = self.num_models
        if not (1 <= n <= self.num_models):
            raise ValueError(f"use_num_models={n} outside 1..{self.num_models}")
        out = AniModel(**{k: getattr(self, k) for k in (
            "species", "Rcr", "Rca", "EtaR", "EtaA", "Zeta", "ShfR", "ShfA", "ShfZ",
            "self_energies", "dims", "celu_alpha")})
        out.weights = self.weights[:n]
        out.repulsion = self.repulsion
        return out


def write_model(path: str, m: AniModel) -> None:
    S, M, L = m.num_species, m.num_models, m.num_layers
    assert all(len(d) == L + 1 for d in m.dims)
    assert all(d[0] == m.aev_len for d in m.dims), "first layer width must equal AEV length"
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<6I", S, M, L, len(m.ShfR), len(m.ShfA), len(m.ShfZ)))
        f.write(struct.pack("<6d", m.Rcr, m.Rca, m.EtaR, m.EtaA, m.Zeta, m.celu_alpha))
        for arr in (m.ShfR, m.ShfA, m.ShfZ):
            f.write(np.asarray(arr, dtype="<f8").tobytes())
        for s in range(S):
            sym = m.species[s].encode()[:4]
            f.write(sym + b"\0" * (4 - len(sym)))
            f.write(struct.pack("<d", float(m.self_energies[s])))
            f.write(struct.pack(f"<{L + 1}I", *m.dims[s]))
        for mi in range(M):
            for s in range(S):
                for l in range(L):
                    W, b = m.weights[mi][s][l]
                    assert W.shape == (m.dims[s][l + 1], m.dims[s][l]) and b.shape == (m.dims[s][l + 1],)
                    f.write(np.ascontiguousarray(W, dtype="<f4").tobytes())
                    f.write(np.ascontiguousarray(b, dtype="<f4").tobytes())
        if m.repulsion is not None:
            f.write(b"REPULXTB")
            f.write(struct.pack("<d", float(m.repulsion["cutoff"])))
            for key in ("y_ab", "sqrt_alpha_ab", "k_rep_ab"):
                a = np.ascontiguousarray(m.repulsion[key], dtype="<f8")
                assert a.shape == (S, S), key
                f.write(a.tobytes())


def read_model(path: str) -> AniModel:
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise ValueError(f"{path}: not an ANIHIP01 model file")
    off = 8
    S, M, L, nR, nA, nZ = struct.unpack_from("<6I", buf, off)
    off += 24
    Rcr, Rca, EtaR, EtaA, Zeta, alpha = struct.unpack_from("<6d", buf, off)
    off += 48

    def arr(n, dt, cnt_bytes):
        nonlocal off
        a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        off += n * cnt_bytes
        return a

    ShfR, ShfA, ShfZ = arr(nR, "<f8", 8), arr(nA, "<f8", 8), arr(nZ, "<f8", 8)
    species, sae, dims = [], [], []
    for _ in range(S):
        species.append(buf[off:off + 4].rstrip(b"\0").decode())
        off += 4
        sae.append(struct.unpack_from("<d", buf, off)[0])
        off += 8
        dims.append(list(struct.unpack_from(f"<{L + 1}I", buf, off)))
        off += 4 * (L + 1)
    weights = []
    for _ in range(M):
        per_s = []
        for s in range(S):
            per_l = []
            for l in range(L):
                o, i = dims[s][l + 1], dims[s][l]
                W = arr(o * i, "<f4", 4).reshape(o, i)
                b = arr(o, "<f4", 4)
                per_l.append((W, b))
            per_s.append(per_l)
        weights.append(per_s)
    rep = None
    if off != len(buf):
        if buf[off:off + 8] != b"REPULXTB" or len(buf) - off != 16 + 3 * 8 * S * S:
            raise ValueError(f"{path}: trailing bytes ({len(buf) - off})")
        off += 8
        cutoff = struct.unpack_from("<d", buf, off)[0]
        off += 8
        rep = {"cutoff": cutoff}
        for key in ("y_ab", "sqrt_alpha_ab", "k_rep_ab"):
            rep[key] = arr(S * S, "<f8", 8).reshape(S, S)
    return AniModel(species, Rcr, Rca, EtaR, EtaA, Zeta, ShfR, ShfA, ShfZ,
                    np.array(sae), dims, weights, alpha, rep)


# --------------------------------------------------------------------------------------------------------
# seeded synthetic weights ("seeded N(0, 1/sqrt(fan_in)) fp32, biases 0" — here a
# variance-matched uniform so that the generator is a few lines of integer arithmetic with no library RNG)
# --------------------------------------------------------------------------------------------------------

def _splitmix64(n: int, seed: int) -> np.ndarray:
    """n 64-bit outputs of the splitmix64 stream started at ``seed`` (vectorised)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform_pm1(n: int, seed: int) -> np.ndarray:
    """n doubles uniform in [-1, 1)."""
    z = _splitmix64(n, seed)
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


# GFN2-xTB repulsion parameters (alpha, effective nuclear charge) by element [RECALL: Bannwarth, Ehlert, Grimme, JCTC 15
# (2019) SI; the values torchani's RepulsionXTB tabulates].  A converted real model carries its own tables; these feed
# the synthetic models and tests.
XTB_REP_ALPHA = {"H": 2.213717, "C": 1.247655, "N": 1.682689, "O": 2.165712, "F": 2.421394, "S": 1.214553, "Cl": 1.577144}
XTB_REP_YEFF = {"H": 1.105388, "C": 4.231078, "N": 5.242592, "O": 5.784415, "F": 7.021486, "S": 14.995090, "Cl": 17.353134}


def xtb_repulsion(species: Sequence[str], cutoff: float = 5.1) -> dict:
    """Tables of ``RepulsionXTB(cutoff=5.1, symbols=..., cutoff_fn="smooth")`` (models/ani_models.py:53):
    y_ab = y_a y_b, sqrt_alpha_ab = sqrt(alpha_a alpha_b), k_rep_ab = 1.5 except 1.0 for H-H."""
    a = np.array([XTB_REP_ALPHA[s] for s in species])
    y = np.array([XTB_REP_YEFF[s] for s in species])
    k = np.full((len(species), len(species)), 1.5)
    for i, si in enumerate(species):
        for j, sj in enumerate(species):
            if si == "H" and sj == "H":
                k[i, j] = 1.0
    return {"cutoff": float(cutoff), "y_ab": np.outer(y, y), "sqrt_alpha_ab": np.sqrt(np.outer(a, a)), "k_rep_ab": k}


ANI2X_SPECIES = ["H", "C", "N", "O", "S", "F", "Cl"]
# Hartree; SURVEY.md §8a row a8 [RECALL]
ANI2X_SAE = [-0.5978583943827134, -38.08933878049795, -54.711968298621066, -75.19106774742086,
             -398.1577125334925, -99.80348506781634, -460.1681939421027]
ANI2X_HIDDEN = {"H": [256, 192, 160], "C": [224, 192, 160], "N": [192, 160, 128], "O": [192, 160, 128],
                "S": [160, 128, 96], "F": [160, 128, 96], "Cl": [160, 128, 96]}

ANI1X_SPECIES = ["H", "C", "N", "O"]
ANI1X_SAE = [-0.600953, -38.08316, -54.707756, -75.194466]
ANI1X_HIDDEN = {"H": [160, 128, 96], "C": [144, 112, 96], "N": [128, 112, 96], "O": [128, 112, 96]}


def _fill_weights(model: AniModel, num_models: int, seed: int, bias_scale: float, out_scale: float = 0.2) -> None:
    model.weights = []
    stream = 0
    for mi in range(num_models):
        per_s = []
        for s in range(model.num_species):
            per_l = []
            for l in range(model.num_layers):
                o, i = model.dims[s][l + 1], model.dims[s][l]
                a = math.sqrt(3.0 / i)  # uniform(-a, a) has variance 1/fan_in
                if l == model.num_layers - 1:
                    # output layer at 1/5 of that: atomic energies ~0.1 Ha and water forces of rms ~15 kcal/mol/A,
                    # the scale of a trained ANI-2x (unit-variance outputs give rms ~80, which would make the
                    # absolute force tolerances of the tests 5x harder than they are for the real model)
                    a *= out_scale
                stream += 1
                W = (_uniform_pm1(o * i, seed * 1000003 + stream) * a).astype(np.float32).reshape(o, i)
                stream += 1
                b = (_uniform_pm1(o, seed * 1000003 + stream) * bias_scale).astype(np.float32)
                per_l.append((W, b))
            per_s.append(per_l)
        model.weights.append(per_s)


def synthetic_model(kind: str = "ani2x", num_models: int = 8, seed: int = 2024,
                    bias_scale: float = 0.05, out_scale: float = 0.2, repulsion: bool = False) -> AniModel:
    """Shape-identical stand-in for a trained model.

    ``kind``: ``"ani2x"`` (7 species, AEV 1008, nets of SURVEY.md §8a row a7), ``"ani1x"`` (4 species,
    AEV 384) or ``"tiny"`` (3 species, AEV 51, small nets — unit-test size).  ``out_scale`` scales the output
    layer (0.2: force magnitudes of a trained ANI-2x on water; the MD stand-in uses 0.02 so that the random energy
    surface, which has no minimum at the starting structure, stays within a few kT).
    """
    if kind == "ani2x":
        sp, sae, hid = ANI2X_SPECIES, ANI2X_SAE, ANI2X_HIDDEN
        m = AniModel(sp, 5.1, 3.5, 19.7, 12.5, 14.1,
                     0.8 + 0.26875 * np.arange(16), np.array([0.8, 1.1375, 1.475, 1.8125, 2.15, 2.4875, 2.825, 3.1625]),
                     (2 * np.arange(4) + 1) * math.pi / 8, np.array(sae), [])
    elif kind == "ani1x":
        sp, sae, hid = ANI1X_SPECIES, ANI1X_SAE, ANI1X_HIDDEN
        m = AniModel(sp, 5.2, 3.5, 16.0, 8.0, 32.0,
                     0.9 + 0.26875 * np.arange(16), np.array([0.9, 1.55, 2.2, 2.85]),
                     (2 * np.arange(8) + 1) * math.pi / 16, np.array(sae), [])
    elif kind == "tiny":
        sp = ["H", "C", "O"]
        sae = [-0.5, -38.0, -75.0]
        hid = {"H": [24, 16, 8], "C": [20, 16, 12], "O": [16, 12, 8]}
        m = AniModel(sp, 5.1, 3.5, 19.7, 12.5, 14.1,
                     0.8 + 0.86 * np.arange(5), np.array([0.8, 1.7, 2.6]),
                     (2 * np.arange(2) + 1) * math.pi / 4, np.array(sae), [])
    else:
        raise ValueError(kind)
    m.dims = [[m.aev_len] + hid[s] + [1] for s in sp]
    _fill_weights(m, num_models, seed, bias_scale, out_scale)
    if repulsion:
        m.repulsion = xtb_repulsion(sp, cutoff=m.Rcr)
    return m
