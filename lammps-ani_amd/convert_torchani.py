"""torchani ``state_dict`` -> flat ``*.anim`` model file.

The reference exports a TorchScript archive of ``LammpsANI(model)`` (``models/ani_models.py:112-122``); its numbers all
live in the ``state_dict`` of the wrapped torchani model (``aev_computer``, ``neural_networks``, ``energy_shifter`` —
the attributes ``models/lammps_ani.py:76-93`` requires).  This converter reads such a mapping (anything with
``.items()`` whose values have ``.numpy()`` or are array-like; ``torch.load(path, map_location="cpu")`` gives one) and
writes the format of :mod:`model_file`.  Run where torchani is installed::

    python -c "import torch, torchani; torch.save(torchani.models.ANI2x().state_dict(), 'ani2x.sd.pt')"
    python -m lammps_ani_amd.convert_torchani ani2x.sd.pt ani2x.anim --species H C N O S F Cl

Key layout understood (torchani's ``Ensemble`` of ``ANIModel``, each atomic network an ``nn.Sequential`` of
``Linear``/``CELU``): ``[<prefix>.]neural_networks.<member>.<symbol>.<2*layer>.{weight,bias}``; a single ``ANIModel``
(no ensemble) omits ``<member>``.  AEV constants are the ``aev_computer`` buffers ``EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ``
(any shape, flattened) plus the two cutoffs, which torchani keeps as Python attributes, not buffers — pass them
(``--rcr/--rca``, defaults 5.1 / 3.5 as in every ``pair_style ani 5.1`` line of the reference).  Self energies are
``energy_shifter.self_energies``.  Untestable here against real weights (torchani is not in the container, SURVEY.md
§8c); ``tests/test_convert_torchani.py`` round-trips a synthetic model through this key layout.
"""
from __future__ import annotations

import argparse
import re
from typing import Dict, List, Mapping, Sequence

import numpy as np

from .model_file import AniModel, write_model


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def _find(sd: Mapping[str, object], suffix: str) -> np.ndarray:
    hits = [k for k in sd if k == suffix or k.endswith("." + suffix)]
    if len(hits) != 1:
        raise KeyError(f"expected exactly one key ending in '{suffix}', found {hits}")
    return _np(sd[hits[0]])


def from_state_dict(sd: Mapping[str, object], species: Sequence[str], rcr: float = 5.1, rca: float = 3.5,
                    celu_alpha: float = 0.1, rep_cutoff: float = None) -> AniModel:
    """Build an :class:`AniModel` from a torchani-style state dict.  ``species`` is the model's species order
    (= LAMMPS type order, ``src/pair_ani.cpp:110``)."""
    species = list(species)
    pat = re.compile(r"(?:^|\.)neural_networks\.(?:(\d+)\.)?([A-Za-z]+)\.(\d+)\.(weight|bias)$")
    nets: Dict[int, Dict[str, Dict[int, Dict[str, np.ndarray]]]] = {}
    for k, v in sd.items():
        mt = pat.search(k)
        if not mt:
            continue
        member = int(mt.group(1)) if mt.group(1) is not None else 0
        sym, idx, kind = mt.group(2), int(mt.group(3)), mt.group(4)
        nets.setdefault(member, {}).setdefault(sym, {}).setdefault(idx, {})[kind] = _np(v)
    if not nets:
        raise KeyError("no 'neural_networks.<member>.<symbol>.<index>.weight' keys found")
    members = sorted(nets)
    if members != list(range(len(members))):
        raise ValueError(f"ensemble members are not 0..M-1: {members}")
    missing = [s for s in species if s not in nets[0]]
    if missing:
        raise KeyError(f"species {missing} have no network in the state dict (has {sorted(nets[0])})")

    shf_r, shf_a, shf_z = (_find(sd, "aev_computer." + n).astype(np.float64).ravel() for n in ("ShfR", "ShfA", "ShfZ"))
    eta_r, eta_a, zeta = (float(_find(sd, "aev_computer." + n).ravel()[0]) for n in ("EtaR", "EtaA", "Zeta"))
    sae = _find(sd, "energy_shifter.self_energies").astype(np.float64).ravel()
    if sae.shape[0] != len(species):
        raise ValueError(f"{sae.shape[0]} self energies for {len(species)} species")

    model = AniModel(species, float(rcr), float(rca), eta_r, eta_a, zeta, shf_r, shf_a, shf_z, sae, [], [], celu_alpha)
    # optional pairwise repulsion (RepulsionXTB buffers; its cutoff is a Python attribute there: pass rep_cutoff)
    rep_keys = [k for k in sd if k.endswith("y_ab")]
    if rep_keys and rep_cutoff is not None:
        pre = rep_keys[0][: -len("y_ab")]
        model.repulsion = {"cutoff": float(rep_cutoff), "y_ab": _np(sd[pre + "y_ab"]).astype(np.float64),
                           "sqrt_alpha_ab": _np(sd[pre + "sqrt_alpha_ab"]).astype(np.float64),
                           "k_rep_ab": _np(sd[pre + "k_rep_ab"]).astype(np.float64)}
    dims: List[List[int]] = []
    for s in species:
        layers = [nets[0][s][i] for i in sorted(nets[0][s])]
        d = [layers[0]["weight"].shape[1]] + [l["weight"].shape[0] for l in layers]
        dims.append(d)
    if any(d[0] != model.aev_len for d in dims):
        raise ValueError(f"first-layer widths {[d[0] for d in dims]} do not match the AEV length {model.aev_len} implied by "
                         f"{len(species)} species and the shift grids")
    if any(d[-1] != 1 for d in dims) or len({len(d) for d in dims}) != 1:
        raise ValueError("every atomic network must end in one output and have the same depth")
    model.dims = dims
    for mi in members:
        per_s = []
        for s in species:
            idxs = sorted(nets[mi][s])
            per_s.append([(np.ascontiguousarray(nets[mi][s][i]["weight"], dtype=np.float32),
                           np.ascontiguousarray(nets[mi][s][i]["bias"], dtype=np.float32)) for i in idxs])
        model.weights.append(per_s)
    return model


def to_state_dict(m: AniModel) -> Dict[str, np.ndarray]:
    """The inverse mapping (numpy values) — what the round-trip test feeds to :func:`from_state_dict`."""
    sd: Dict[str, np.ndarray] = {
        "aev_computer.EtaR": np.array([m.EtaR]), "aev_computer.ShfR": np.asarray(m.ShfR).reshape(1, -1),
        "aev_computer.EtaA": np.array([m.EtaA]), "aev_computer.Zeta": np.array([m.Zeta]),
        "aev_computer.ShfA": np.asarray(m.ShfA).reshape(1, 1, -1, 1), "aev_computer.ShfZ": np.asarray(m.ShfZ).reshape(1, 1, 1, -1),
        "energy_shifter.self_energies": np.asarray(m.self_energies),
    }
    if m.repulsion is not None:
        for key in ("y_ab", "sqrt_alpha_ab", "k_rep_ab"):
            sd["rep_calc." + key] = np.asarray(m.repulsion[key])
    for mi in range(m.num_models):
        for si, sym in enumerate(m.species):
            for li, (W, b) in enumerate(m.weights[mi][si]):
                sd[f"neural_networks.{mi}.{sym}.{2 * li}.weight"] = W
                sd[f"neural_networks.{mi}.{sym}.{2 * li}.bias"] = b
    return sd


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("state_dict", help="file readable by torch.load (a state_dict)")
    ap.add_argument("out", help="output .anim file")
    ap.add_argument("--species", nargs="+", required=True, help="species symbols in model (= LAMMPS type) order")
    ap.add_argument("--rcr", type=float, default=5.1)
    ap.add_argument("--rca", type=float, default=3.5)
    ap.add_argument("--celu-alpha", type=float, default=0.1)
    ap.add_argument("--rep-cutoff", type=float, default=None,
                    help="also convert the RepulsionXTB tables found in the state dict, with this cutoff (5.1 in the reference)")
    a = ap.parse_args(argv)
    import torch
    sd = torch.load(a.state_dict, map_location="cpu")
    if hasattr(sd, "state_dict"):
        sd = sd.state_dict()
    m = from_state_dict(sd, a.species, a.rcr, a.rca, a.celu_alpha, a.rep_cutoff)
    write_model(a.out, m)
    print(f"wrote {a.out}: {m.num_species} species, {m.num_models} members, AEV {m.aev_len}, dims {m.dims[0]}")


if __name__ == "__main__":
    main()
