"""NeuroChem model directory -> flat ``*.anim`` model file (second half).

The reference builds its ANI-1xnr models from NeuroChem resources (``models/ani_models.py:38-46``:
``torchani.neurochem.load_model_from_info_file`` on ``external/ani-1xnr/model/ani-1xnr.info`` -- an empty submodule in
the reference tree, so no file of this kind is at hand).  The layout below is restated FROM MEMORY of torchani's
``neurochem`` reader [RECALL]; it is exercised here only on files written by :func:`write_neurochem` in this module
(``tests/test_convert_neurochem.py``), i.e. it is as "unpinned" as the rest until run on a real model directory.

    <name>.info          four lines: constants file, self-energy file, ensemble prefix, ensemble size
    *.params             text, ``key = value`` lines: Rcr, Rca, EtaR=[..], ShfR=[..], Zeta=[..], ShfZ=[..], EtaA=[..],
                         ShfA=[..], Atyp=[H,C,N,O]
    sae*.dat             lines ``H,0=-0.600952980000`` (symbol, index = Hartree); ``#`` comments
    <prefix><i>/networks/ANN-<symbol>.nnf
                         network description: ``inputsize=384;`` and ``layer [ nodes=160; activation=9; type=0;
                         weights=ANN-H-l1.wparam; biases=ANN-H-l1.bparam; ]`` blocks, either as plain text or as bz2 data
                         behind a header that ends in ``=`` and one more byte; activation 9 = CELU(0.1), 6 = linear
    *.wparam / *.bparam  raw little-endian float32, weights ``[nodes][inputs]`` row-major, biases ``[nodes]``

Usage::

    python -m lammps_ani_amd.convert_neurochem external/ani-1xnr/model/ani-1xnr.info ani1xnr.anim
"""
from __future__ import annotations

import argparse
import bz2
import os
import re
from typing import Dict, List, Tuple

import numpy as np

from .model_file import AniModel, write_model

ACT_CELU, ACT_LINEAR = 9, 6


def _read_params(path: str) -> Dict[str, object]:
    out: Dict[str, object] = {}
    for line in open(path):
        line = line.split("#")[0].strip()
        if "=" not in line:
            continue
        k, v = (t.strip() for t in line.split("=", 1))
        if v.startswith("["):
            items = [t.strip() for t in v.strip("[]").split(",") if t.strip()]
            out[k] = items if k == "Atyp" else np.array([float(t) for t in items])
        else:
            try:
                out[k] = float(v)
            except ValueError:
                out[k] = v
    need = ("Rcr", "Rca", "EtaR", "ShfR", "Zeta", "ShfZ", "EtaA", "ShfA", "Atyp")
    missing = [k for k in need if k not in out]
    if missing:
        raise KeyError(f"{path}: missing {missing}")
    return out


def _read_sae(path: str, species: List[str]) -> np.ndarray:
    sae = {}
    for line in open(path):
        line = line.split("#")[0].strip()
        mt = re.match(r"^([A-Za-z]+)\s*,\s*(\d+)\s*=\s*([-+0-9.eE]+)$", line)
        if mt:
            sae[mt.group(1)] = float(mt.group(3))
    missing = [s for s in species if s not in sae]
    if missing:
        raise KeyError(f"{path}: no self energy for {missing}")
    return np.array([sae[s] for s in species], dtype=np.float64)


def _nnf_text(path: str) -> str:
    raw = open(path, "rb").read()
    if b"layer" in raw and b"inputsize" in raw:      # plain text
        return raw.decode("ascii", "replace")
    i = raw.find(b"=")                                 # compressed: header ... '=' + one byte, then bz2
    if i < 0:
        raise ValueError(f"{path}: neither a text network description nor the compressed form")
    return bz2.decompress(raw[i + 2:]).decode("ascii", "replace")


def _read_network(nnf: str) -> Tuple[int, List[Tuple[np.ndarray, np.ndarray]], List[int]]:
    text = _nnf_text(nnf)
    mt = re.search(r"inputsize\s*=\s*(\d+)\s*;", text)
    if not mt:
        raise ValueError(f"{nnf}: no inputsize")
    n_in = int(mt.group(1))
    here = os.path.dirname(nnf)
    layers, acts = [], []
    prev = n_in
    for block in re.findall(r"layer\s*\[(.*?)\]", text, flags=re.S):
        kv = dict((k.strip(), v.strip()) for k, v in re.findall(r"([A-Za-z_]+)\s*=\s*([^;]+);", block))
        nodes, act = int(kv["nodes"]), int(kv["activation"])
        W = np.fromfile(os.path.join(here, kv["weights"]), dtype="<f4")
        b = np.fromfile(os.path.join(here, kv["biases"]), dtype="<f4")
        if W.size != nodes * prev or b.size != nodes:
            raise ValueError(f"{nnf}: layer of {nodes} nodes on {prev} inputs, but {kv['weights']} holds {W.size} and "
                             f"{kv['biases']} {b.size} values")
        layers.append((np.ascontiguousarray(W.reshape(nodes, prev)), np.ascontiguousarray(b)))
        acts.append(act)
        prev = nodes
    if not layers:
        raise ValueError(f"{nnf}: no layers")
    if acts[-1] != ACT_LINEAR or any(a != ACT_CELU for a in acts[:-1]):
        raise ValueError(f"{nnf}: activations {acts}; this build evaluates CELU(0.1) hidden layers (code {ACT_CELU}) and a "
                         f"linear output (code {ACT_LINEAR}) only")
    return n_in, layers, acts


def from_info_file(info: str) -> AniModel:
    base = os.path.dirname(os.path.abspath(info))
    lines = [l.strip() for l in open(info) if l.strip()]
    if len(lines) < 4:
        raise ValueError(f"{info}: expected four lines (constants, self energies, ensemble prefix, ensemble size)")
    consts, saef, prefix, nens = (os.path.join(base, lines[0]), os.path.join(base, lines[1]), os.path.join(base, lines[2]),
                                  int(lines[3]))
    c = _read_params(consts)
    species = list(c["Atyp"])
    model = AniModel(species, float(c["Rcr"]), float(c["Rca"]), float(c["EtaR"][0]), float(c["EtaA"][0]), float(c["Zeta"][0]),
                     np.asarray(c["ShfR"], np.float64), np.asarray(c["ShfA"], np.float64), np.asarray(c["ShfZ"], np.float64),
                     _read_sae(saef, species), [], [], 0.1)
    for mi in range(nens):
        per_s = []
        for s in species:
            n_in, layers, _ = _read_network(os.path.join(f"{prefix}{mi}", "networks", f"ANN-{s}.nnf"))
            if n_in != model.aev_len:
                raise ValueError(f"network of {s} in member {mi} takes {n_in} inputs; the constants imply an AEV of {model.aev_len}")
            per_s.append(layers)
        model.weights.append(per_s)
    model.dims = [[model.aev_len] + [W.shape[0] for W, _ in layers] for layers in model.weights[0]]
    if len({len(d) for d in model.dims}) != 1 or any(d[-1] != 1 for d in model.dims):
        raise ValueError("every atomic network must end in one output and have the same depth")
    return model


def write_neurochem(m: AniModel, root: str, name: str = "model", compressed: bool = False) -> str:
    """The inverse (what the round-trip test feeds to :func:`from_info_file`); returns the path of the .info file."""
    os.makedirs(root, exist_ok=True)

    def arr(v):
        return "[" + ",".join(f"{float(x):.17e}" for x in np.asarray(v).ravel()) + "]"

    with open(os.path.join(root, f"{name}.params"), "w") as f:
        f.write("TM = 1\n")
        f.write(f"Rcr = {m.Rcr:.17e}\nRca = {m.Rca:.17e}\n")
        f.write(f"EtaR = {arr([m.EtaR])}\nShfR = {arr(m.ShfR)}\nZeta = {arr([m.Zeta])}\nShfZ = {arr(m.ShfZ)}\n")
        f.write(f"EtaA = {arr([m.EtaA])}\nShfA = {arr(m.ShfA)}\nAtyp = [{','.join(m.species)}]\n")
    with open(os.path.join(root, "sae.dat"), "w") as f:
        for i, s in enumerate(m.species):
            f.write(f"{s},{i}={float(m.self_energies[i]):.17e}\n")
    for mi in range(m.num_models):
        d = os.path.join(root, f"train{mi}", "networks")
        os.makedirs(d, exist_ok=True)
        for si, s in enumerate(m.species):
            text = f"inputsize={m.aev_len};\n"
            nl = len(m.weights[mi][si])
            for li, (W, b) in enumerate(m.weights[mi][si]):
                wn, bn = f"ANN-{s}-l{li + 1}.wparam", f"ANN-{s}-l{li + 1}.bparam"
                np.asarray(W, "<f4").tofile(os.path.join(d, wn))
                np.asarray(b, "<f4").tofile(os.path.join(d, bn))
                act = ACT_LINEAR if li == nl - 1 else ACT_CELU
                text += f"layer [\n  nodes={W.shape[0]};\n  activation={act};\n  type=0;\n  weights={wn};\n  biases={bn};\n]\n"
            with open(os.path.join(d, f"ANN-{s}.nnf"), "wb") as f:
                f.write(b"NNF1 compressed =\n" + bz2.compress(text.encode("ascii")) if compressed else text.encode("ascii"))
    info = os.path.join(root, f"{name}.info")
    with open(info, "w") as f:
        f.write(f"{name}.params\nsae.dat\ntrain\n{m.num_models}\n")
    return info


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("info", help="NeuroChem .info file")
    ap.add_argument("out", help="output .anim file")
    a = ap.parse_args(argv)
    m = from_info_file(a.info)
    write_model(a.out, m)
    print(f"wrote {a.out}: species {m.species}, {m.num_models} members, AEV {m.aev_len}, dims {m.dims[0]}")


if __name__ == "__main__":
    main()
