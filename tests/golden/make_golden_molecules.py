#!/usr/bin/env python3
"""The reference's own test molecules as numeric fixtures.

Run in the authoring container only (needs /root/reference and oracle/_ref):

    python tests/golden/make_golden_molecules.py

Inputs (data files held by the reference's tests, read in place, nothing of them is copied as text):
  /root/reference/src/pytorch/molecules/{1hvj,1hvk,2iuz,3hkw,3hky,3lka,3o99}_ligand.mol2
      -- the seven ligands of TestSymmetryFunctions.py:37-70 (test_compare_with_native)
  /root/reference/src/pytorch/molecules/water.pdb
      -- the 306-atom water box, 15 A cubic cell, of TestSymmetryFunctions.py:72-105
         (test_compare_waterbox_pbc_with_native)

For every system: positions (A), ANI-2x species, cell; and what the REFERENCE CPU implementation
(oracle/_ref/libnnpops_ref.so = CpuANISymmetryFunctions compiled in place) returns for them with the
ANI-2x symmetry functions: the full radial / angular AEV and the position gradient of
sum(w_r * radial) + sum(w_a * angular) for the deterministic weights of weights() below.

Output: tests/golden/molecules_ref.npz (compressed; the AEV arrays are mostly zeros).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
MOLECULES = "/root/reference/src/pytorch/molecules"
LIGANDS = ["1hvj", "1hvk", "2iuz", "3hkw", "3hky", "3lka", "3o99"]
ANI2X_SPECIES = {"H": 0, "C": 1, "N": 2, "O": 3, "S": 4, "F": 5, "Cl": 6}


def element_of(name):
    """Element from a mol2 / pdb atom name ('C12', 'CAA', 'H5', 'Cl1'): two-letter halogens when spelt with a
    lower-case second letter, otherwise the first letter (every atom of these files is H C N O S F)."""
    if len(name) > 1 and name[:2] in ("Cl", "Br"):
        return name[:2]
    return name[0].upper()


def read_mol2(path):
    pos, elements, on = [], [], False
    for line in open(path):
        if line.startswith("@<TRIPOS>"):
            on = line.strip() == "@<TRIPOS>ATOM"
            continue
        f = line.split()
        if on and len(f) >= 6:
            elements.append(element_of(f[1]))
            pos.append([float(f[2]), float(f[3]), float(f[4])])
    return np.array(pos, np.float32), elements


def read_pdb(path):
    pos, elements, cell = [], [], None
    for line in open(path):
        if line.startswith("CRYST1"):
            a, b, c = float(line[6:15]), float(line[15:24]), float(line[24:33])
            assert [float(line[33:40]), float(line[40:47]), float(line[47:54])] == [90.0, 90.0, 90.0]
            cell = np.diag([a, b, c]).astype(np.float32)
        elif line.startswith(("ATOM", "HETATM")):
            pos.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
            elements.append(line[76:78].strip().capitalize())
        elif line.startswith("ENDMDL"):
            break
    return np.array(pos, np.float32), elements, cell


def weights(shape, k):
    """Deterministic upstream gradients on a 1/64 grid (so that a last-bit difference between two libms' cos() cannot
    change them): the test regenerates them with this same formula instead of storing them."""
    i, j = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    return (np.round(np.cos(0.37 * i + 1.3 * j + 0.5 + k) * 64) / 64).astype(np.float32)


def main():
    from nnpops_amd import workloads
    from oracle import RefAni, have_ref
    assert have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rf, af = workloads.ani2x_functions()
    out = {"names": np.array(LIGANDS + ["water"])}
    systems = []
    for name in LIGANDS:
        pos, el = read_mol2(os.path.join(MOLECULES, f"{name}_ligand.mol2"))
        systems.append((name, pos, el, None))
    pos, el, cell = read_pdb(os.path.join(MOLECULES, "water.pdb"))
    assert len(el) == 306 and cell is not None
    systems.append(("water", pos, el, cell))
    for k, (name, pos, el, cell) in enumerate(systems):
        species = np.array([ANI2X_SPECIES[e] for e in el], np.int32)
        ref = RefAni(7, 5.1, 3.5, species, rf, af, periodic=cell is not None)
        radial, angular = ref.forward(pos, cell)
        wr, wa = weights(radial.shape, k), weights(angular.shape, k + 100)
        out[f"{name}_positions"] = pos
        out[f"{name}_species"] = species
        if cell is not None:
            out[f"{name}_cell"] = cell
        out[f"{name}_radial"], out[f"{name}_angular"] = radial, angular
        out[f"{name}_grad"] = ref.backward(wr, wa)
        counts = {e: el.count(e) for e in sorted(set(el))}
        print(f"{name:6s} {len(el):4d} atoms {counts}  sum(aev) = {radial.sum() + angular.sum():.4f}  max|grad| = {np.abs(out[f'{name}_grad']).max():.4f}")
    path = os.path.join(HERE, "molecules_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
