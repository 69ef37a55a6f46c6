"""In-kernel timeline of the encoder-backward launch INSIDE the training step (the last chain kernel of a step), EMX_CHAIN_TIMING
build:   python tools/abl_bwd_instep.py build_ab/libemx_timing.so"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import encodermap_amd as em
lib = _lib.load()
rng = np.random.default_rng(0)
data = rng.normal(0, 1, (20000, 128)).astype(np.float32)
p = em.Parameters(periodicity=float("inf"), batch_size=4096, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=100, seed=7)
e = em.EncoderMap(p, data, read_only=True)
e._engine.run_steps(41)   # (5 x 8-step graph + 1: the last launch sequence is a whole step)
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.int64)
lib.emx_debug_read.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read(buf.ctypes.data) == 0
for blk in (0, 1):
    clk = buf[blk * 2048: blk * 2048 + 512].reshape(8, 64)
    for w in (0, 7):
        c = clk[w] - clk[w, 0]
        print(f"block {blk*100} wave {w}:  stage loads issued {c[1]}  staged + barrier {c[2]}  loop end {c[60]}  end {c[61]}")
        for l in range(2):
            s3, s4, s5, s6, s7 = (c[3 + 5 * l + i] for i in range(5))
            prev = c[2] if l == 0 else c[7 + 5 * (l - 1)]
            print(f"   u{l}: start {prev:6d}  head {s3-prev:5d}  mfma+requests {s4-s3:5d}  records {s5-s4:5d}  epilogue {s6-s5:5d}  barrier {s7-s6:5d}")
