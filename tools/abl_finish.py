"""In-kernel timeline of finish_step_kernel inside the captured training step (EMX_CHAIN_TIMING build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import encodermap_amd as em
lib = _lib.load()
rng = np.random.default_rng(0)
data = rng.standard_normal((100000, 128)).astype(np.float32)
p = em.Parameters(periodicity=float("inf"), batch_size=4096, n_steps=50, dist_sig_parameters=(4.5, 12, 6, 1, 2, 6))
e = em.EncoderMap(p, data, read_only=True)
e._engine.run_steps(50)
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.int64)
lib.emx_debug_read_fin.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read_fin(buf.ctypes.data) == 0
print("finish_step stamps (clk since entry):", (buf[:6] - buf[0]).tolist())
print("adam block 0 stamps [entry, grads summed, w/m/v stored, mirror written, reg partial written]:", (buf[16:21] - buf[16]).tolist())
