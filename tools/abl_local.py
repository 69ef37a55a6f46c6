"""Timeline (100 MHz wall clock) of emx_mlp_wgrad_adam_local on an EMX_CHAIN_TIMING build: every block workgroup, every gather
workgroup, the scalar-epilogue workgroup.   python tools/abl_local.py build_ab/libemx_timing.so [B D]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import encodermap_amd as em
lib = _lib.load()
B, D = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 128)
rng = np.random.default_rng(0)
data = rng.normal(0, 1, (20000, D)).astype(np.float32)
p = em.Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=100, seed=7)
e = em.EncoderMap(p, data, read_only=True)
np.set_printoptions(linewidth=220, precision=1, suppress=True)
lay = e._engine.layout
widths = [lay.net_dim] + [l.n_out for l in lay.layers]
n_blocks = sum(((a + 31) // 32) * ((b + 31) // 32) for a, b in zip(widths[:-1], widths[1:]))
for rep in range(3):
    e._engine.run_steps(40)
    torch.cuda.synchronize()
    buf = np.zeros(8192, dtype=np.int64)
    lib.emx_debug_read.argtypes = [ctypes.c_void_p]
    assert lib.emx_debug_read(buf.ctypes.data) == 0
    t = buf[:8 * 1001].reshape(1001, 8).astype(np.float64)
    w = t[:512]
    w = w[(w[:, 5] > w[:, 0]) & (w[:, 5] - w[:, 0] < 1e4) & (w[:, 2] > w[:, 0])]
    t0 = w[:, 0].min()
    f = lambda x: np.where(x > 0, (x - t0) / 100.0, np.nan)
    w, g, fin = f(w), f(t[512:1000]), f(t[1000])
    g = g[~np.isnan(g[:, 0])]
    print("local ran:", e._engine._tail_ran, " blocks:", n_blocks, " gather workgroups:", len(g))
    for k, n in enumerate(("entry", "-", "products done", "shares in LDS", "optimizer issued", "exit", "block summed", "slab done")):
        if k == 1:
            continue
        c = w[:, k]
        print(f"  {n:28s} min {np.nanmin(c):6.2f}  median {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c, 90):6.2f}  max {np.nanmax(c):6.2f}")
    if rep == 2:
        print("  products done, by block:"); print(w[:, 2])
    if len(g):
        print(f"  gather: entry min {np.nanmin(g[:, 0]):6.2f} max {np.nanmax(g[:, 0]):6.2f}  done median {np.nanmedian(g[:, 1]):6.2f}  max {np.nanmax(g[:, 1]):6.2f}")
    print(f"  epilogue: entry {fin[0]:6.2f}  sums done {fin[1]:6.2f}  counters advanced {fin[2]:6.2f}")
