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
w = buf[32:44].astype(np.float64)
print("adam block 0 (100 MHz wall clock, us since its entry): first slab's operands in %.2f  updated %.2f  operand store written %.2f (these three: the LAST slab of the block) | all items done %.2f  end %.2f" % tuple((w[k] - w[0]) / 100 for k in (1, 2, 3, 4, 5)))
print("first gather block: entry %+.2f us after adam block 0, lasts %.2f us;  last gather block: entry %+.2f, lasts %.2f" % ((w[8] - w[0]) / 100, (w[9] - w[8]) / 100, (w[10] - w[0]) / 100, (w[11] - w[10]) / 100))

print("adam block 0: layer record in registers (operand loads can be issued) %.2f us after entry" % ((w[6] - w[0]) / 100))
