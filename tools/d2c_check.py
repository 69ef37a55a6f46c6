import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from encodermap_amd import _lib as L
from oracle import reference_ops as R
from tests.test_backmap_gpu import make_chain, dev, t64, PI
for B, A in [(40, 60), (8, 228), (4, 457)]:
    rng = np.random.default_rng(2)
    lengths, angles, dih = make_chain(rng, B, A)
    ml = lengths.mean(0)
    plane = torch.empty(B, A, 3, device="cuda"); out = torch.empty(B, A, 3, device="cuda")
    L.chain_in_plane_fwd(dev(ml), dev(angles), plane)
    L.dihedrals_to_cartesian_fwd(dev(dih), PI, plane, out)
    po = R.chain_in_plane(t64(ml.astype(np.float64).astype(np.float32))[None, :], t64(angles))
    xo = R.dihedrals_to_cartesian(t64(dih) + PI, t64(plane.cpu().numpy())).numpy()
    x = out.cpu().numpy().astype(np.float64)
    err = np.abs(x - xo).max(axis=(0, 2))
    bl = np.linalg.norm(np.diff(x, axis=1), axis=2)
    blo = np.linalg.norm(np.diff(xo, axis=1), axis=2)
    print(B, A, "max xyz err", err.max(), "bond err", np.abs(bl - blo).max(), "extent", np.abs(xo).max())
    print("  err by atom:", np.array2string(err, precision=1, max_line_width=200))
