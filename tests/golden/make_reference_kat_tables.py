"""Extracts the literal known-answer TABLES (data, not code) that the reference's own tests hold
and stores them as small fixtures.  Run in the build container only (needs /root/reference):

    python tests/golden/make_reference_kat_tables.py

Sources (relative to AG-Peter/encodermap @ 2025-05-23):
  * tests/test_dihedral_to_cartesian.py:104-148  33x3 helix table (dihedrals [phi,psi,0]x10, straight chain of 33)
  * tests/test_dihedral_to_cartesian.py:186-197  7x3 straight_tetrahedral_chain(bond_lengths=[1,2,3,1,2,3])
  * tests/data/known_angles_{dists,angles,dihedrals}.npy + known_angles.pdb coordinates (6 frames x 4 atoms)
  * tests/test_angles.py:72-103  the 28 backbone angles (rad) of tests/data/Ala10_helix.pdb, with the N/CA/C coordinates
    of that file (30 atoms) as the input
  * encodermap/parameters/parameters.py:611-639, 794-828  the two ``_defaults`` tables (keys and literal values) of
    ``Parameters`` and ``ADCParameters``, evaluated from the AST of the assignments (``dict(Parameters._defaults, **dict(...))``
    is resolved by hand; pi and inf are the only names)
The tables are parsed out of the test file with a regex over numeric literals; no reference code is executed
or copied.
"""
import ast
import json
import math
import os
import re

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def numbers_between(text, start_marker, end_marker):
    i = text.index(start_marker)
    j = text.index(end_marker, i)
    return [float(t) for t in re.findall(r"-?\d+\.\d*(?:e-?\d+)?", text[i:j])]


def _literal(node, base=None):
    """value of a literal expression that may use ``pi`` / ``np.pi`` / ``float("inf")``, arithmetic on them, and
    ``Parameters._defaults[key]`` (looked up in ``base``)"""
    if isinstance(node, ast.Subscript) and getattr(node.value, "attr", "") == "_defaults":
        return base[_literal(node.slice)]
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, (ast.List, ast.Tuple)):
        vals = [_literal(e) for e in node.elts]
        return vals if isinstance(node, ast.List) else {"__tuple__": vals}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        return -_literal(node.operand)
    if isinstance(node, ast.BinOp):
        a, b = _literal(node.left), _literal(node.right)
        return {ast.Mult: a * b, ast.Div: a / b if b else None, ast.Add: a + b, ast.Sub: a - b}[type(node.op)]
    if isinstance(node, ast.Name) and node.id == "pi":
        return math.pi
    if isinstance(node, ast.Attribute) and node.attr == "pi":
        return math.pi
    if isinstance(node, ast.Call) and getattr(node.func, "id", "") == "float":
        return {"__float__": _literal(node.args[0])}      # float("inf"): kept symbolic (JSON has no inf)
    raise ValueError(ast.dump(node))


def parameter_defaults():
    """{"Parameters": {...}, "ADCParameters": {...}} -- the latter already merged over the former, as the reference's
    ``dict(Parameters._defaults, **dict(...))`` does."""
    tree = ast.parse(open(os.path.join(REF, "encodermap/parameters/parameters.py")).read())
    found = {}
    for cls in [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name in ("Parameters", "ADCParameters")]:
        for st in cls.body:
            if isinstance(st, ast.Assign) and any(getattr(t, "id", "") == "_defaults" for t in st.targets):
                call = st.value
                assert isinstance(call, ast.Call) and call.func.id == "dict"
                kw = call.keywords
                if call.args:      # dict(Parameters._defaults, **dict(k=v, ...))
                    inner = [k.value for k in kw if k.arg is None]
                    assert len(inner) == 1 and inner[0].func.id == "dict"
                    kw = inner[0].keywords
                found[cls.name] = {k.arg: _literal(k.value, found.get("Parameters")) for k in kw}
    found["ADCParameters"] = dict(found["Parameters"], **found["ADCParameters"])
    return found


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
    out["parameter_defaults"] = parameter_defaults()
    with open(os.path.join(HERE, "reference_kat_tables.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("helix", helix.shape, "chain", chain.shape, "frames", np.array(frames).shape)


if __name__ == "__main__":
    main()
