#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.npz).  Runs in the build container only.

The reference's arithmetic for this path lives in torchani, which is not installed and not vendored
(SURVEY.md §8c), and its own golden vectors need trained ANI-2x weights that are not available offline.
This script is therefore NOT the reference: it is an independent restatement, in plain torch fp64 with
autograd, of the same published algorithm, organised the way the reference's python path is organised:

  * half pair list ``atom_index12`` + ``diff_vector = x[i] - x[j]`` + ``distances``
    (src/ani_csrc/ani.cpp:143-144, models/lammps_ani.py:156-166),
  * pyaev-style AEV from those pairs with ``index_add`` (radial for both ends of each pair, angular from
    triples of close pairs around a central atom) — call site models/lammps_ani.py:294-296,
  * ensemble of per-species CELU(0.1) MLPs, mean over members, + self energies, ghosts masked with
    species -1 (models/lammps_ani.py:218-233, src/ani_csrc/ani.cpp:226-228),
  * force = -autograd.grad(E, coordinates), virial = -sym(dE/d(diff)^T @ diff) (models/lammps_ani.py:195-216),
  * Hartree -> kcal/mol (src/ani_csrc/ani.h:9).

It shares no code with oracle/ani_oracle.c (which uses hand-derived analytic gradients over per-centre
neighbour lists), so agreement between the two to ~1e-10 is a meaningful check of both.

Fixtures hold inputs (positions, types, box, the exact neighbour lists used) and expected outputs; model
weights are regenerated from (kind, num_models, seed) at test time and guarded by a CRC of the model file.
"""
import math
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

_pkg.load()
from lammps_ani_amd import harness as hx  # noqa: E402
from lammps_ani_amd import model_file as mf  # noqa: E402

HARTREE2KCALMOL = 627.5094738898777
torch.set_default_dtype(torch.float64)


def cutoff_cosine(d, rc):
    return 0.5 * torch.cos(d * (math.pi / rc)) + 0.5


def triples(a12: np.ndarray, ntotal: int):
    """For every central atom, every unordered pair of its incident (close) pairs: (central, pairA, pairB, signA, signB).
    sign = +1 if the central atom is the first end of the pair (vector from centre = -diff), -1 otherwise."""
    inc = [[] for _ in range(ntotal)]
    for p in range(a12.shape[1]):
        i, j = int(a12[0, p]), int(a12[1, p])
        inc[i].append((p, -1.0))  # centre i: x_j - x_i = -(x_i - x_j)
        inc[j].append((p, +1.0))  # centre j: x_i - x_j = +diff
    c, pa, pb, sa, sb = [], [], [], [], []
    for atom, lst in enumerate(inc):
        for u in range(len(lst)):
            for v in range(u + 1, len(lst)):
                c.append(atom)
                pa.append(lst[u][0]); sa.append(lst[u][1])
                pb.append(lst[v][0]); sb.append(lst[v][1])
    return (np.array(c, dtype=np.int64), np.array(pa, dtype=np.int64), np.array(pb, dtype=np.int64),
            np.array(sa), np.array(sb))


def torch_reference(model: mf.AniModel, species: np.ndarray, x: np.ndarray, nlocal: int, a12: np.ndarray, compat: bool):
    S, nR, nA, nZ = model.num_species, len(model.ShfR), len(model.ShfA), len(model.ShfZ)
    ntotal = len(species)
    sp = torch.from_numpy(species.astype(np.int64))
    coords = torch.from_numpy(x.astype(np.float64)).clone().requires_grad_(True)
    a12t = torch.from_numpy(a12.astype(np.int64))
    diff = coords.index_select(0, a12t[0]) - coords.index_select(0, a12t[1])
    dist = diff.norm(2, -1)
    ShfR, ShfA, ShfZ = (torch.from_numpy(np.asarray(v, dtype=np.float64)) for v in (model.ShfR, model.ShfA, model.ShfZ))

    # radial
    if compat:
        sel = torch.arange(dist.shape[0])
    else:
        sel = (dist <= model.Rcr).nonzero().flatten()
    d_r = dist.index_select(0, sel)
    a12_r = a12t.index_select(1, sel)
    terms = 0.25 * torch.exp(-model.EtaR * (d_r[:, None] - ShfR[None, :]) ** 2) * cutoff_cosine(d_r, model.Rcr)[:, None]
    radial = torch.zeros(ntotal * S, nR)
    sp12 = sp[a12_r]
    index12 = a12_r * S + sp12.flip(0)
    radial = radial.index_add(0, index12[0], terms)
    radial = radial.index_add(0, index12[1], terms)
    radial = radial.reshape(ntotal, S * nR)

    # angular
    close = (dist <= model.Rca).nonzero().flatten()
    a12_c = a12t.index_select(1, close).numpy()
    c, pa, pb, sa, sb = triples(a12_c, ntotal)
    npair_types = S * (S + 1) // 2
    angular = torch.zeros(ntotal * npair_types, nA * nZ)
    if len(c):
        diff_c = diff.index_select(0, close)
        vA = diff_c.index_select(0, torch.from_numpy(pa)) * torch.from_numpy(sa)[:, None]
        vB = diff_c.index_select(0, torch.from_numpy(pb)) * torch.from_numpy(sb)[:, None]
        dA, dB = vA.norm(2, -1), vB.norm(2, -1)
        cos_angles = (vA * vB).sum(-1) / torch.clamp(dA * dB, min=1e-10)
        angles = torch.acos(0.95 * cos_angles)
        fcj12 = cutoff_cosine(dA, model.Rca) * cutoff_cosine(dB, model.Rca)
        factor1 = ((1 + torch.cos(angles[:, None, None] - ShfZ[None, None, :])) / 2) ** model.Zeta
        factor2 = torch.exp(-model.EtaA * ((dA + dB)[:, None, None] / 2 - ShfA[None, :, None]) ** 2)
        ang_terms = (2 * factor1 * factor2 * fcj12[:, None, None]).reshape(-1, nA * nZ)
        # species of the two non-central ends
        a12_cn = a12_c
        endA = np.where(sa < 0, a12_cn[1, pa], a12_cn[0, pa])
        endB = np.where(sb < 0, a12_cn[1, pb], a12_cn[0, pb])
        s1, s2 = species[endA], species[endB]
        lo, hi = np.minimum(s1, s2), np.maximum(s1, s2)
        triu = lo * S - lo * (lo - 1) // 2 + (hi - lo)  # row-major upper triangle incl. diagonal
        index = torch.from_numpy(c * npair_types + triu)
        angular = angular.index_add(0, index, ang_terms)
    angular = angular.reshape(ntotal, npair_types * nA * nZ)
    aev = torch.cat([radial, angular], dim=-1)

    # networks: ghosts (index >= nlocal) are padding (species -1)
    sp_pad = sp.clone()
    sp_pad[nlocal:] = -1
    atomic = torch.zeros(ntotal)
    for s in range(S):
        idx = (sp_pad == s).nonzero().flatten()
        if idx.numel() == 0:
            continue
        xin = aev.index_select(0, idx)
        acc = torch.zeros(idx.numel())
        for m in range(model.num_models):
            h = xin
            L = model.num_layers
            for l in range(L):
                W = torch.from_numpy(model.weights[m][s][l][0].astype(np.float64))
                b = torch.from_numpy(model.weights[m][s][l][1].astype(np.float64))
                h = h @ W.t() + b
                if l < L - 1:
                    # CELU written out: torch 2.10's fp64 celu backward rounds 1/alpha to fp32 (1.5e-8 relative
                    # error per layer, measured), which would cap fixture accuracy at ~1e-6 kcal/mol/A
                    al = model.celu_alpha
                    h = torch.where(h > 0, h, al * (torch.exp(torch.clamp(h, max=0.0) / al) - 1))
            acc = acc + h.flatten()
        atomic = atomic.index_add(0, idx, acc / model.num_models + float(model.self_energies[s]))
    energy = atomic.sum()
    if model.repulsion is not None:
        # pairwise repulsion on the half list (models/lammps_ani.py:186-193,300-330; RepulsionXTB with the "smooth" cutoff,
        # atomic units inside): a pair with a ghost end counts half (ghost_flags), pairs of two ghosts are not in the list
        rep = model.repulsion
        i12, j12 = a12t[0], a12t[1]
        inside = (dist < rep["cutoff"]).nonzero().flatten()
        rr = dist.index_select(0, inside)
        si, sj = sp.index_select(0, i12.index_select(0, inside)), sp.index_select(0, j12.index_select(0, inside))
        y = torch.from_numpy(np.asarray(rep["y_ab"], dtype=np.float64))[si, sj]
        sa = torch.from_numpy(np.asarray(rep["sqrt_alpha_ab"], dtype=np.float64))[si, sj]
        kk = torch.from_numpy(np.asarray(rep["k_rep_ab"], dtype=np.float64))[si, sj]
        d_bohr = rr * 1.8897261258369282
        fc = torch.exp(1.0 - 1.0 / (1.0 - (rr / rep["cutoff"]) ** 2).clamp(min=1e-10))
        w = torch.where((i12.index_select(0, inside) < nlocal) & (j12.index_select(0, inside) < nlocal),
                        torch.ones_like(rr), 0.5 * torch.ones_like(rr))
        energy = energy + (w * y / d_bohr * torch.exp(-sa * d_bohr ** kk) * fc).sum()
    gx, gdiff = torch.autograd.grad([energy], [coords, diff])
    virial = gdiff.t() @ diff
    virial = -(virial.t() + virial) / 2
    return dict(energy=energy.item() * HARTREE2KCALMOL,
                force=(-gx).numpy() * HARTREE2KCALMOL,
                eatom=atomic[:nlocal].detach().numpy() * HARTREE2KCALMOL,
                virial=virial.detach().numpy() * HARTREE2KCALMOL,
                aev=aev[:nlocal].detach().numpy())


def model_crc(model: mf.AniModel) -> int:
    p = "/tmp/_golden_model.anim"
    mf.write_model(p, model)
    with open(p, "rb") as f:
        return zlib.crc32(f.read())


CASES = [
    # name, system factory, model kind, M, seed, cutoff
    ("water30_pbc_ani2x_m8", lambda: hx.read_lammps_data(os.path.join(ROOT, "tests/golden/water-0.8nm.data")), "ani2x", 8, 2024),
    ("water30_open_ani2x_m8", lambda: _open(hx.read_lammps_data(os.path.join(ROOT, "tests/golden/water-0.8nm.data"))), "ani2x", 8, 2024),
    ("mixed64_pbc_ani1x_m2", lambda: hx.random_box(64, 4, 9.0, seed=11), "ani1x", 2, 7),
    ("mixed40_pbc_tiny_m3", lambda: hx.random_box(40, 3, 8.0, seed=5), "tiny", 3, 3),
    ("mixed96_pbc_ani2x_m2", lambda: hx.random_box(96, 7, 10.5, seed=23), "ani2x", 2, 99),
    # with the optional pairwise repulsion block
    ("mixed64_pbc_ani1x_m2_rep", lambda: hx.random_box(64, 4, 9.0, seed=11), "ani1x", 2, 7, True),
]


def _open(s):
    s.periodic = (False, False, False)
    return s


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, factory, kind, M, seed, *opt in CASES:
        rep_flag = bool(opt and opt[0])
        sys_ = factory()
        model = mf.synthetic_model(kind, M, seed, repulsion=rep_flag)
        full = hx.decompose(sys_, cutoff=model.Rcr, skin=2.0, half=False)
        half = hx.decompose(sys_, cutoff=model.Rcr, skin=2.0, half=True)
        a12 = half.atom_index12().reshape(2, -1)
        rec = dict(kind=kind, num_models=M, seed=seed, repulsion=int(rep_flag), model_crc=model_crc(model),
                   sys_x=sys_.x, sys_types=sys_.types, boxlo=sys_.boxlo, boxhi=sys_.boxhi,
                   periodic=np.array(sys_.periodic, dtype=np.int32), cutoff=model.Rcr, skin=2.0,
                   nlocal=full.nlocal, x=full.x, types=full.types, numneigh=full.numneigh, jlist=full.jlist,
                   owner_lidx=full.owner_lidx, half_numneigh=half.numneigh, half_jlist=half.jlist)
        assert np.array_equal(full.x, half.x)
        for compat in (False, True):
            r = torch_reference(model, full.species, full.x, full.nlocal, a12, compat)
            tagc = "compat" if compat else "strict"
            for k, v in r.items():
                rec[f"{tagc}_{k}"] = v
            print(f"{name:28s} {tagc:6s} E={r['energy']:.6f} |F|max={np.abs(r['force']).max():.4f} ntotal={full.ntotal} npairs={full.npairs}")
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)


if __name__ == "__main__":
    main()
