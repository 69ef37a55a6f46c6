"""Timeline (100 MHz wall clock) of emx_mlp_wgrad_adam on an EMX_CHAIN_TIMING build: every weight-gradient workgroup, every gather
workgroup, the scalar-epilogue workgroup.   python tools/abl_tail.py build_ab/libemx_timing.so [B D]"""
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
for rep in range(3):
    e._engine.run_steps(40)
    torch.cuda.synchronize()
    buf = np.zeros(8192, dtype=np.int64)
    lib.emx_debug_read.argtypes = [ctypes.c_void_p]
    assert lib.emx_debug_read(buf.ctypes.data) == 0
    t = buf[:8 * 1001].reshape(1001, 8).astype(np.float64)
    live = t[:, 0] > 0
    t0 = t[live, 0].min()
    t = np.where(t > 0, (t - t0) / 100.0, np.nan)
    n_wg = 6 * _lib.mlp_wgrad_slices(B)      # (the default network: six output blocks)
    w = t[:n_wg]
    g = t[512:1000][live[512:1000]]
    print("tail ran:", e._engine._tail_ran, " weight-gradient workgroups:", len(w), " gather workgroups:", len(g))
    names = ("entry", "partial stores issued", "drained", "all arrived", "optimizer issued", "drained")
    for k, n in enumerate(names):
        c = w[:, k]
        print(f"  {n:24s} min {np.nanmin(c):6.2f}  median {np.nanmedian(c):6.2f}  p90 {np.nanpercentile(c, 90):6.2f}  max {np.nanmax(c):6.2f}")
    if rep == 2:
        print("  partial stores issued, by workgroup (rows: slices, columns: output blocks):")
        nt = 6
        print(w[:, 1].reshape(-1, nt).T)
        print("  optimizer time by workgroup:")
        print((w[:, 4] - w[:, 3]).reshape(-1, nt).T)
    if len(g):
        print(f"  gather: entry max {np.nanmax(g[:, 0]):6.2f}  done median {np.nanmedian(g[:, 1]):6.2f}  max {np.nanmax(g[:, 1]):6.2f}")
    print(f"  epilogue: entry {t[1000, 0]:6.2f}  done {t[1000, 1]:6.2f}")
