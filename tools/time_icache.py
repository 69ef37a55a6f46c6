"""The fused dense-layer launch of the C2 step behind (a) itself, (b) a launch that does nothing, (c) a launch that walks 56 KB of
straight-line code on every CU (evicting the instruction caches): how much of the launch's "first touch" inside the step is code?
    python tools/time_icache.py      (needs tools/libicache_probe.so: see tools/icache_probe.hip)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, encodermap_amd as em

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libicache_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cfg = bench.CONFIGS["C2"]
p = em.Parameters(n_steps=2000, seed=7, **cfg["kw"])
B = cfg["kw"]["batch_size"]
e = em.EncoderMap(p, np.zeros((B, cfg["D"]), "float32"), read_only=True, device="cuda:0", use_graph=False)
eng = e._engine
eng.set_data(bench.make_data(cfg, "cuda:0", 200000))
eng.run_steps(20)
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
N = 300
def probe(kb, wgs=512):
    assert lib.probe_launch(kb, wgs, st) == 0
from encodermap_amd import _lib
lay = eng.layout
def bwd():
    _lib.mlp_backward(eng.plan_lean, 0, lay.n_enc, eng.dlat, None, lay.n_enc - 1, 0, B, latent=eng._sk.grad)
def sk():
    eng._latent_costs()
def tail():
    eng._wgrad()
SIZES = (0, 8, 16, 24, 32, 40, 48, 56)
t_p = {kb: bench._time_launches(lambda: probe(kb), N) * 1e6 for kb in SIZES}
print("the launch behind a launch that walks .. KB of straight-line code on every CU (pair minus that launch alone), us:")
print("                         " + "".join(f"{kb:>8d}" for kb in SIZES) + "   alone, back to back")
for fn, nm in ((eng._forward, "fused dense layers"), (sk, "all-pairs"), (bwd, "encoder backward"), (tail, "weight gradients + optimizer")):
    t_a = bench._time_launches(fn, N) * 1e6
    row = []
    for kb in SIZES:
        t_b = bench._time_launches(lambda: (probe(kb), fn()), N) * 1e6
        row.append(t_b - t_p[kb])
    print(f"{nm:<25s}" + "".join(f"{v:8.2f}" for v in row) + f"   {t_a:8.2f}")
