"""Times emx_allatom_backmap on a synthetic ~470-atom molecule, next to the oracle's
restatement of the reference's serial loop on the host.   python tools/time_allatom.py [n_frames]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import encodermap_amd as em  # noqa: E402
from encodermap_amd import _lib as L  # noqa: E402
from molecules import random_molecule  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
xyz, bonds, cq, sq = random_molecule(160, 7)
top = em.AllAtomTopology(xyz, bonds, cq, sq)
A, J = len(xyz), top.table.n_torsions
rng = np.random.default_rng(0)
tg = rng.uniform(-np.pi, np.pi, (n, J)).astype(np.float32)
dev = torch.device("cuda:0")
q, m = top.table.device_tables(dev)
x0 = torch.from_numpy(xyz).to(dev)
t = torch.from_numpy(tg).to(dev)
o = torch.empty(n, A, 3, device=dev)
for _ in range(3):
    L.allatom_backmap(x0, q, m, t, o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    L.allatom_backmap(x0, q, m, t, o)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
far_atoms = sum(len(f) for f in top.table.far_sides)
print(f"A={A} J={J} frames={n}: {ms:.3f} ms/launch, {n / ms * 1e3:.3e} frames/s, {n * far_atoms / ms * 1e3 / 1e9:.2f} G atom-rotations/s, "
      f"output {n * A * 12 / ms * 1e3 / 1e9:.1f} GB/s")
if "--cpu" in sys.argv:
    from oracle import allatom_ops as O

    far = [O.far_side(A, bonds, qq[1:3]) for qq in top.table.quadruplets]
    k = 8
    t0 = time.perf_counter()
    O.allatom_backmap(xyz, top.table.quadruplets, far, tg[:k])
    dt = time.perf_counter() - t0
    print(f"oracle (numpy restatement of the serial loop, 1 core): {k / dt:.2f} frames/s")
