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

# ---- the fused frame passes of the ADC step (emx_adc_frame_fwd / emx_adc_frame_bwd), wave 0 of workgroup 0
C_ = len(range(A)[1::3]); P_ = C_ * (C_ - 1) // 2; ldo = (P_ + 3) // 4 * 4
Y = torch.randn(B, 2 * (2 * A - 5), device="cuda")
cart = xyz + 0.05 * torch.randn(B, A, 3, device="cuda")
angp, dihp = torch.empty(B, A - 2, device="cuda"), torch.empty(B, A - 3, device="cuda")
ip, op = torch.empty(B, ldo, device="cuda"), torch.empty(B, ldo, device="cuda")
sw = torch.zeros(L.sketchmap_split_bytes(B, P_), dtype=torch.uint8, device="cuda")
ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda"); sc = torch.ones(1, device="cuda")
for _ in range(3):
    L.adc_frame_fwd(Y, 2 * np.pi, (0, A - 2), None, angp, (2 * (A - 2), 2 * (A - 2) + A - 3), dihp, ln, np.pi, plane, xyz, cart, 1, 3, C_, ip, op, split_ws=sw)
    L.adc_frame_bwd(op, ip, "mean_abs", 0.125, sc, ws, xyz, dihp, np.pi, plane, 1, 3, C_, dd, dp)
torch.cuda.synchronize()
buf2 = (ctypes.c_longlong * 32)()
lib.emx_debug_read_af.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read_af(buf2) == 0
t2 = np.array(buf2[:], dtype=np.int64)
print("frame forward : " + "  ".join(f"{n} {int(t2[k] - t2[0])}" for k, n in enumerate(["entry", "atan2 + staged", "chain done (wave 0)", "barrier", "d2c done", "pairs done", "split done"])))
print("frame backward: " + "  ".join(f"{n} {int(t2[16 + k] - t2[16])}" for k, n in enumerate(["entry", "cost done", "pair-distance backward done", "barrier", "d2c backward done"])))
