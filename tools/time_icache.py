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
def probe(big, wgs=512):
    assert lib.probe_launch(big, wgs, st) == 0
def both(big):
    probe(big); eng._forward()
t_f = bench._time_launches(eng._forward, N) * 1e6
for big, name in ((0, "a launch that does nothing"), (1, "a launch that walks 56 KB of code on every CU")):
    t_p = bench._time_launches(lambda: probe(big), N) * 1e6
    t_b = bench._time_launches(lambda: both(big), N) * 1e6
    print(f"fused launch behind {name}: {t_b - t_p:.2f} us  (pair {t_b:.2f}, the other launch alone {t_p:.2f}; fused alone back to back {t_f:.2f})")
# the same for the encoder's backward launch and the all-pairs launch
from encodermap_amd import _lib
lay = eng.layout
def bwd():
    _lib.mlp_backward(eng.plan_lean, 0, lay.n_enc, eng.dlat, None, lay.n_enc - 1, 0, B, latent=eng._sk.grad)
def sk():
    eng._latent_costs()
for fn, nm in ((bwd, "encoder backward"), (sk, "all-pairs")):
    t_a = bench._time_launches(fn, N) * 1e6
    out = [f"{nm}: alone {t_a:.2f} us"]
    for big in (0, 1):
        t_p = bench._time_launches(lambda: probe(big), N) * 1e6
        t_b = bench._time_launches(lambda: (probe(big), fn()), N) * 1e6
        out.append(f"behind {'big code' if big else 'nothing'} {t_b - t_p:.2f}")
    print(", ".join(out))
