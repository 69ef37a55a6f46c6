"""Extracts the literal known-answer TABLES (data, not code) that the reference's own tests hold
and stores them as small fixtures.  Run in the build container only (needs /root/reference):

    python tests/golden/make_reference_kat_tables.py

Sources (relative to AG-Peter/encodermap @ 2025-05-23):
  * tests/test_dihedral_to_cartesian.py:104-148  33x3 helix table (dihedrals [phi,psi,0]x10, straight chain of 33)
  * tests/test_dihedral_to_cartesian.py:186-197  7x3 straight_tetrahedral_chain(bond_lengths=[1,2,3,1,2,3])
  * tests/data/known_angles_{dists,angles,dihedrals}.npy + known_angles.pdb coordinates (6 frames x 4 atoms)
  * tests/test_angles.py:72-103  the 28 backbone angles (rad) of tests/data/Ala10_helix.pdb, with the N/CA/C coordinates
    of that file (30 atoms) as the input
The tables are parsed out of the test file with a regex over numeric literals; no reference code is executed
or copied.
"""
import json
import os
import re

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def numbers_between(text, start_marker, end_marker):
    i = text.index(start_marker)
    j = text.index(end_marker, i)
    return [float(t) for t in re.findall(r"-?\d+\.\d*(?:e-?\d+)?", text[i:j])]


def main():
    src = open(os.path.join(REF, "tests/test_dihedral_to_cartesian.py")).read()
    helix = numbers_between(src, "result = np.array(\n            [\n                [\n                    [", "* 10,")
    helix = np.array(helix, dtype=np.float64).reshape(33, 3)
    chain = numbers_between(src, "def test_straight_tetrahedral_chain_with_bond_lenght", "cartesian = em_tf1.straight_tetrahedral_chain")
    chain = np.array(chain, dtype=np.float64).reshape(7, 3)
    out = {
        "helix_33x3": helix.tolist(),
        "helix_phi_deg": 57.8, "helix_psi_deg": 47.0,
        "tetrahedral_chain_7x3": chain.tolist(),
        "tetrahedral_chain_bond_lengths": [1, 2, 3, 1, 2, 3],
    }
    # known_angles fixture: coordinates from the PDB text + the three answer arrays
    frames, cur = [], []
    for line in open(os.path.join(REF, "tests/data/known_angles.pdb")):
        if line.startswith(("ATOM", "HETATM")):
            cur.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
        elif line.startswith("ENDMDL") and cur:
            frames.append(cur)
            cur = []
    if cur:
        frames.append(cur)
    out["known_angles_xyz_angstrom"] = frames
    for k in ("dists", "angles", "dihedrals"):
        out[f"known_angles_{k}"] = np.load(os.path.join(REF, f"tests/data/known_angles_{k}.npy")).tolist()
    # Ala10 helix: backbone N, CA, C coordinates in residue order + the literal angle table of tests/test_angles.py
    bb = [[float(l[30:38]), float(l[38:46]), float(l[46:54])] for l in open(os.path.join(REF, "tests/data/Ala10_helix.pdb"))
          if l.startswith("ATOM") and l[12:16].strip() in ("N", "CA", "C")]
    src_a = open(os.path.join(REF, "tests/test_angles.py")).read()
    ala = numbers_between(src_a, "moldata.angles,", "def test_chain_in_plane")
    assert len(bb) == 30 and len(ala) == 28, (len(bb), len(ala))
    out["ala10_backbone_xyz_angstrom"] = bb
    out["ala10_backbone_angles_rad"] = ala
    with open(os.path.join(HERE, "reference_kat_tables.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("helix", helix.shape, "chain", chain.shape, "frames", np.array(frames).shape)


if __name__ == "__main__":
    main()
