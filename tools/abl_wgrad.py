"""Timeline (100 MHz wall clock) of the weight-gradient kernel on an EMX_CHAIN_TIMING build: workgroups (block 0, slice 0),
(block 0, last slice), the scalar-epilogue workgroup.   python tools/abl_wgrad.py build_ab/libemx_timing.so"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import encodermap_amd as em
lib = _lib.load()
B, D = 4096, 128
rng = np.random.default_rng(0)
data = rng.normal(0, 1, (20000, D)).astype(np.float32)
p = em.Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=100, seed=7)
e = em.EncoderMap(p, data, read_only=True)
e._engine.run_steps(40)
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.int64)
lib.emx_debug_read.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read(buf.ctypes.data) == 0
t = buf[4096:4096 + 64].reshape(4, 16)
t0 = min(int(r[0]) for r in t if r[0])
for name, r in zip(("block 0 slice 0", "block 0 last slice", "epilogue wg", "(last column, last slice)"), t):
    print(name, "entry %.2f us; since entry: loads issued %.2f  products done %.2f  folded %.2f  stores issued %.2f  | epilogue done %.2f" % (
        (r[0] - t0) / 100.0, *[((r[k] - r[0]) / 100.0 if r[k] else float("nan")) for k in (1, 2, 3, 4, 5)]),
        "| layer record in registers %.2f" % ((r[6] - r[0]) / 100.0 if r[6] else float("nan")))
