"""Per-workgroup timeline of the all-pairs tile kernel (EMX_CHAIN_TIMING build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.load()
B, D = 4096, 128
g = torch.Generator(device="cuda"); g.manual_seed(0)
cen = torch.randn(16, D, generator=g, device="cuda") * 4
X = cen[torch.randint(0, 16, (B,), generator=g, device="cuda")] + torch.randn(B, D, generator=g, device="cuda")
Z = torch.randn(B, 2, generator=g, device="cuda")
ws = torch.zeros(_lib.sketchmap_ws_bytes(B, 2) // 4, device="cuda")
split = None
if len(sys.argv) > 2 and sys.argv[2] == "split":
    split = torch.zeros(_lib.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
    _lib.sketchmap_split(X, split)
for _ in range(5): _lib.sketchmap_tiles(X, Z, float("inf"), (16.0, 12, 6, 1, 2, 6), 0, B, ws, split)
torch.cuda.synchronize()
buf = np.zeros(128 + 3 * 4096, dtype=np.int64)
lib.emx_debug_read_sk.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read_sk(buf.ctypes.data) == 0
t0 = buf[0]
for w in range(4):
    c = buf[w * 32: w * 32 + 32]
    if split is not None:
        print(f"block {512*w}: since entry: first loads issued {c[1]-c[0]}", " | ".join(f"c{k}: staged {c[2+2*k]-c[0]} mfma issued {c[3+2*k]-c[0]}" for k in range(4)),
              f"| K loop done+barrier {c[20]-c[0]}  pairs done {c[21]-c[0]}  block_sum {c[22]-c[0]}  end {c[23]-c[0]}")
        continue
    print(f"block {512*w}: entry at {c[0]-t0:7d}; since entry: staged0 {c[1]-c[0]}",
          " | ".join(f"chunk{k}: ready {c[2+2*k]-c[0]} mfma-done {c[3+2*k]-c[0]}" for k in range(4)),
          f"| norms+barrier {c[20]-c[0]}  pairs done {c[21]-c[0]}  block_sum {c[22]-c[0]}  end {c[23]-c[0]}")

n = 2176                      # (the XCD-aware grid: 8 x 16 x 17 workgroups, 96 of which leave at once without a stamp)
st = buf[128:128 + 3 * n:3].astype(np.float64); en = buf[129:129 + 3 * n:3].astype(np.float64)
live = (st > 0) & (en > 0)
xcd = np.arange(n) % 8
print("workgroups with a tile:", int(live.sum()), " per XCD:", [int((live & (xcd == x)).sum()) for x in range(8)])
t0 = st[live].min()
for x in range(8):
    m = live & (xcd == x)
    print(f"XCD {x}: last start {((st[m] - t0) * 10e-3).max():6.2f} us, starts later than 3 us: {int((((st[m] - t0) * 10e-3) > 3).sum()):3d}, "
          f"median end {np.median((en[m] - t0) * 10e-3):6.2f}, 90 % end {np.percentile((en[m] - t0) * 10e-3, 90):6.2f}, last end {((en[m] - t0) * 10e-3).max():6.2f}")
st, en = st[live], en[live]
st = (st - t0) * 10e-3; en = (en - t0) * 10e-3   # us at 100 MHz
order = np.argsort(st)
print("kernel span us:", en.max(), " first-wave starts (us) pct 0/50/90/99/100:", np.percentile(st, [0, 50, 90, 99, 100]).round(2))
print("block lifetimes us pct 0/10/50/90/100:", np.percentile(en - st, [0, 10, 50, 90, 100]).round(2))
print("end times us pct 10/50/90/99/100:", np.percentile(en, [10, 50, 90, 99, 100]).round(2))
late = st > 5
print("blocks starting later than 5 us:", int(late.sum()), "their start/end median:", np.median(st[late]).round(2) if late.any() else None, np.median(en[late]).round(2) if late.any() else None)
