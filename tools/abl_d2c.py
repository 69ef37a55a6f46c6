"""Per-phase timeline of the back-mapping frame kernels (workgroup 0), from a -DEMX_D2C_TIMING build:
EMX_LIB_PATH=build_ab/lib_d2ct.so python tools/abl_d2c.py [atoms]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from encodermap_amd import _lib as L
A = int(sys.argv[1]) if len(sys.argv) > 1 else 228
B = 256
rng = np.random.default_rng(0)
dih = torch.from_numpy(rng.uniform(-3, 3, (B, A - 3)).astype(np.float32)).cuda()
ang = torch.from_numpy(rng.normal(2.0, 0.05, (B, A - 2)).astype(np.float32)).cuda()
ln = torch.from_numpy(rng.normal(1.4, 0.02, (A - 1,)).astype(np.float32)).cuda()
plane = torch.empty(B, A, 3, device="cuda"); xyz = torch.empty(B, A, 3, device="cuda")
L.chain_in_plane_fwd(ln, ang, plane)
g = torch.randn(B, A, 3, device="cuda"); dd = torch.empty(B, A - 3, device="cuda"); dp = torch.empty(B, A, 3, device="cuda")
for _ in range(3):
    L.dihedrals_to_cartesian_fwd(dih, np.pi, plane, xyz)
    L.dihedrals_to_cartesian_bwd(dih, np.pi, plane, g, dd, dp)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib = L.load()
lib.emx_debug_read_d2c.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read_d2c(buf) == 0
t = np.array(buf[:], dtype=np.int64)
names = ["entry", "staged", "A done", "B done", "B barrier", "C done", "end"]
for label, off, waves in (("forward", 0, 2), ("backward", 32, 4)):
    for w in range(waves):
        v = t[off + 8 * w: off + 8 * w + 7]
        print(f"{label} wave {w}: " + "  ".join(f"{n} {int(x - v[0])}" for n, x in zip(names, v) if x))
