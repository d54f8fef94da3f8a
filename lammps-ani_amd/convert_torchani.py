"""torchani ``state_dict`` -> flat ``*.anim`` model file.

The reference exports a TorchScript archive of ``LammpsANI(model)`` (``models/ani_models.py:112-122``); its numbers all
live in the ``state_dict`` of the wrapped torchani model (``aev_computer``, ``neural_networks``, ``energy_shifter`` —
the attributes ``models/lammps_ani.py:76-93`` requires).  This converter reads such a mapping (anything with
``.items()`` whose values have ``.numpy()`` or are array-like; ``torch.load(path, map_location="cpu")`` gives one) and
writes the format of :mod:`model_file`.  Run where torchani is installed::

    python -c "import torch, torchani; torch.save(torchani.models.ANI2x().state_dict(), 'ani2x.sd.pt')"
    python -m lammps_ani_amd.convert_torchani ani2x.sd.pt ani2x.anim --species H C N O S F Cl

Key layout: none is assumed.  The reference builds its models with the torchani 2.x API (``ANI2x(neighborlist=...,
strategy=...)``, ``torchani.nn.ANINetworks`` / ``BmmEnsemble``, ``torchani.neurochem``; models/ani_models.py:5-7,17,45,
models/lammps_ani.py:6) whose module names are not in the reference tree, and older torchani releases spell the same
tensors differently, so the networks are DISCOVERED from the tensors themselves:

  * every 2-D ``*.weight`` with a 1-D ``*.bias`` beside it is an affine layer;
  * the path component that is one of the model's element symbols names the species;
  * the last integer component in front of the symbol is the ensemble member (none: a single model);
  * what follows the symbol orders the layers (integers ascending; a component called ``final_layer`` / ``output`` or a
    layer with one output row goes last), and consecutive layers must chain (in_features == previous out_features).

So ``neural_networks.3.H.4.weight`` (torchani <= 2.2), ``neural_networks.members.3.atomics.H.layers.2.weight`` +
``...atomics.H.final_layer.weight`` and ``potentials.nnp.neural_networks.H.0.weight`` all convert.  AEV constants are
found by name wherever they sit — ``EtaR ShfR EtaA Zeta ShfA ShfZ`` or, under a ``radial`` / ``angular`` component,
``eta  shifts  zeta  sections`` (angle sections = ShfZ) — and cutoffs stored as buffers (``...radial.cutoff``,
``...angular.cutoff``, ``Rcr``, ``Rca``) override ``--rcr/--rca`` (defaults 5.1 / 3.5 as in every ``pair_style ani 5.1``
line of the reference).  Self energies: the tensor named ``self_energies`` (an ``energy_shifter`` one is preferred).
When something cannot be identified the error lists every key of the state dict with its shape.

Untestable here against real weights (torchani is not in the container); ``tests/test_convert_torchani.py``
feeds the converter three spellings of the same synthetic model and requires byte-identical model files.
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


def _describe(sd: Mapping[str, object]) -> str:
    return "state dict keys:\n" + "\n".join(f"  {k}  {tuple(_np(v).shape)}" for k, v in sorted(sd.items(), key=lambda kv: kv[0]))


_FINAL_NAMES = ("final_layer", "final", "output", "out", "last")


def discover_networks(sd: Mapping[str, object], species: Sequence[str]):
    """{member: {symbol: [(W, b), ...] in layer order}} from the tensors alone (see the module docstring)."""
    symbols = set(species)
    found: Dict[int, Dict[str, list]] = {}
    for k, v in sd.items():
        if not k.endswith(".weight"):
            continue
        W = _np(v)
        bkey = k[: -len("weight")] + "bias"
        if W.ndim != 2 or bkey not in sd or _np(sd[bkey]).ndim != 1 or _np(sd[bkey]).shape[0] != W.shape[0]:
            continue
        parts = k.split(".")[:-1]
        sym_pos = [i for i, c in enumerate(parts) if c in symbols]
        if not sym_pos:
            continue
        sp = sym_pos[-1]
        ints_before = [int(c) for c in parts[:sp] if c.isdigit()]
        member = ints_before[-1] if ints_before else 0
        tail = parts[sp + 1:]
        is_final = any(c.lower() in _FINAL_NAMES for c in tail) or W.shape[0] == 1
        idx = [int(c) for c in tail if c.isdigit()]
        order = (1 if is_final else 0, idx[-1] if idx else 0, k)
        found.setdefault(member, {}).setdefault(parts[sp], []).append((order, W, _np(sd[bkey])))
    nets = {}
    for mi, per in found.items():
        nets[mi] = {}
        for sym, layers in per.items():
            layers.sort(key=lambda t: t[0])
            for (_, Wp, _), (o, Wn, _) in zip(layers, layers[1:]):
                if Wn.shape[1] != Wp.shape[0]:
                    raise ValueError(f"layers of species {sym}, member {mi} do not chain at {o[2]}: "
                                     f"{Wn.shape[1]} inputs after {Wp.shape[0]} outputs\n" + _describe(sd))
            nets[mi][sym] = [(W, b) for _, W, b in layers]
    return nets


def _canon(name: str) -> str:
    return name.lower().replace("_", "")


def discover_aev(sd: Mapping[str, object]):
    """AEV constants by name, wherever they sit in the module tree.  Returns a dict with whichever of
    EtaR ShfR EtaA Zeta ShfA ShfZ Rcr Rca were found."""
    out = {}
    direct = {"etar": "EtaR", "shfr": "ShfR", "etaa": "EtaA", "zeta": "Zeta", "shfa": "ShfA", "shfz": "ShfZ", "rcr": "Rcr", "rca": "Rca"}
    by_term = {("radial", "eta"): "EtaR", ("radial", "shifts"): "ShfR", ("radial", "cutoff"): "Rcr",
               ("angular", "eta"): "EtaA", ("angular", "zeta"): "Zeta", ("angular", "shifts"): "ShfA",
               ("angular", "sections"): "ShfZ", ("angular", "anglesections"): "ShfZ", ("angular", "cutoff"): "Rca"}
    for k, v in sd.items():
        parts = [_canon(c) for c in k.split(".")]
        last = parts[-1]
        name = direct.get(last)
        if name is None:
            term = "radial" if any("radial" in c for c in parts[:-1]) else ("angular" if any("angular" in c for c in parts[:-1]) else None)
            name = by_term.get((term, last)) if term else None
        if name is None:
            continue
        if name in out and not np.array_equal(np.asarray(out[name]).ravel(), _np(v).astype(np.float64).ravel()):
            raise KeyError(f"two different tensors claim to be {name} (second: {k})\n" + _describe(sd))
        out[name] = _np(v).astype(np.float64)
    return out


def from_state_dict(sd: Mapping[str, object], species: Sequence[str], rcr: float = 5.1, rca: float = 3.5,
                    celu_alpha: float = 0.1, rep_cutoff: float = None) -> AniModel:
    """Build an :class:`AniModel` from a torchani state dict of any release.  ``species`` is the model's species order
    (= LAMMPS type order, ``src/pair_ani.cpp:110``)."""
    species = list(species)
    nets = discover_networks(sd, species)
    if not nets:
        raise KeyError("no atomic networks found: no 2-D '*.weight' (+ '*.bias') whose path names one of the species "
                       f"{species}; e.g. 'neural_networks.<member>.<symbol>.<index>.weight'\n" + _describe(sd))
    members = sorted(nets)
    if members != list(range(len(members))):
        raise ValueError(f"ensemble members are not 0..M-1: {members}\n" + _describe(sd))
    for mi in members:
        missing = [s for s in species if s not in nets[mi]]
        if missing:
            raise KeyError(f"species {missing} have no network in member {mi} (has {sorted(nets[mi])})\n" + _describe(sd))

    aev = discover_aev(sd)
    lacking = [n for n in ("EtaR", "ShfR", "EtaA", "Zeta", "ShfA", "ShfZ") if n not in aev]
    if lacking:
        raise KeyError(f"AEV constants {lacking} not found (looked for EtaR/ShfR/EtaA/Zeta/ShfA/ShfZ and for "
                       "radial.{eta,shifts} / angular.{eta,zeta,shifts,sections})\n" + _describe(sd))
    shf_r, shf_a, shf_z = (aev[n].ravel() for n in ("ShfR", "ShfA", "ShfZ"))
    eta_r, eta_a, zeta = (float(aev[n].ravel()[0]) for n in ("EtaR", "EtaA", "Zeta"))
    rcr = float(aev["Rcr"].ravel()[0]) if "Rcr" in aev else float(rcr)   # a cutoff stored in the file wins over the argument
    rca = float(aev["Rca"].ravel()[0]) if "Rca" in aev else float(rca)
    sae_keys = [k for k in sd if k.split(".")[-1] == "self_energies"]
    pref = [k for k in sae_keys if "energy_shifter" in k] or sae_keys
    if len(pref) != 1:
        raise KeyError(f"expected one 'self_energies' tensor, found {sae_keys}\n" + _describe(sd))
    sae = _np(sd[pref[0]]).astype(np.float64).ravel()
    if sae.shape[0] != len(species):
        raise ValueError(f"{sae.shape[0]} self energies for {len(species)} species")

    model = AniModel(species, rcr, rca, eta_r, eta_a, zeta, shf_r, shf_a, shf_z, sae, [], [], celu_alpha)
    # optional pairwise repulsion (RepulsionXTB buffers; its cutoff is a Python attribute there: pass rep_cutoff)
    rep_keys = [k for k in sd if k.endswith("y_ab")]
    if rep_keys and rep_cutoff is not None:
        pre = rep_keys[0][: -len("y_ab")]
        model.repulsion = {"cutoff": float(rep_cutoff), "y_ab": _np(sd[pre + "y_ab"]).astype(np.float64),
                           "sqrt_alpha_ab": _np(sd[pre + "sqrt_alpha_ab"]).astype(np.float64),
                           "k_rep_ab": _np(sd[pre + "k_rep_ab"]).astype(np.float64)}
    dims: List[List[int]] = []
    for s in species:
        layers = nets[0][s]
        dims.append([layers[0][0].shape[1]] + [W.shape[0] for W, _ in layers])
    if any(d[0] != model.aev_len for d in dims):
        raise ValueError(f"first-layer widths {[d[0] for d in dims]} do not match the AEV length {model.aev_len} implied by "
                         f"{len(species)} species and the shift grids")
    if any(d[-1] != 1 for d in dims) or len({len(d) for d in dims}) != 1:
        raise ValueError("every atomic network must end in one output and have the same depth")
    model.dims = dims
    for mi in members:
        per_s = []
        for si, s in enumerate(species):
            layers = nets[mi][s]
            if [layers[0][0].shape[1]] + [W.shape[0] for W, _ in layers] != dims[si]:
                raise ValueError(f"member {mi}, species {s}: layer widths differ from member 0")
            per_s.append([(np.ascontiguousarray(W, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)) for W, b in layers])
        model.weights.append(per_s)
    return model


def to_state_dict(m: AniModel, layout: str = "legacy") -> Dict[str, np.ndarray]:
    """The inverse mapping (numpy values) in one of the spellings :func:`from_state_dict` must accept — test input.
    "legacy": ``neural_networks.<m>.<sym>.<2l>.weight`` + ``aev_computer.EtaR ...``;  "terms": ensemble members /
    atomics / layers / final_layer modules and radial / angular term buffers, cutoffs stored as buffers."""
    if layout == "legacy":
        sd: Dict[str, np.ndarray] = {
            "aev_computer.EtaR": np.array([m.EtaR]), "aev_computer.ShfR": np.asarray(m.ShfR).reshape(1, -1),
            "aev_computer.EtaA": np.array([m.EtaA]), "aev_computer.Zeta": np.array([m.Zeta]),
            "aev_computer.ShfA": np.asarray(m.ShfA).reshape(1, 1, -1, 1), "aev_computer.ShfZ": np.asarray(m.ShfZ).reshape(1, 1, 1, -1),
            "energy_shifter.self_energies": np.asarray(m.self_energies),
        }
    elif layout == "terms":
        sd = {
            "aev_computer.radial.eta": np.array(m.EtaR), "aev_computer.radial.shifts": np.asarray(m.ShfR),
            "aev_computer.radial.cutoff": np.array(m.Rcr), "aev_computer.angular.cutoff": np.array(m.Rca),
            "aev_computer.angular.eta": np.array(m.EtaA), "aev_computer.angular.zeta": np.array(m.Zeta),
            "aev_computer.angular.shifts": np.asarray(m.ShfA), "aev_computer.angular.sections": np.asarray(m.ShfZ),
            "energy_shifter.self_energies": np.asarray(m.self_energies),
        }
    else:
        raise ValueError(layout)
    if m.repulsion is not None:
        for key in ("y_ab", "sqrt_alpha_ab", "k_rep_ab"):
            sd["rep_calc." + key] = np.asarray(m.repulsion[key])
    for mi in range(m.num_models):
        for si, sym in enumerate(m.species):
            nl = len(m.weights[mi][si])
            for li, (W, b) in enumerate(m.weights[mi][si]):
                if layout == "legacy":
                    stem = f"neural_networks.{mi}.{sym}.{2 * li}"
                else:
                    stem = f"neural_networks.members.{mi}.atomics.{sym}." + ("final_layer" if li == nl - 1 else f"layers.{li}")
                sd[stem + ".weight"] = W
                sd[stem + ".bias"] = b
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
