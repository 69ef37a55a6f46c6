"""Step time of EncoderMap models on either side of the LDS row-block limit (~2400 source columns): periodic data of D features enter
and leave the network 2 D wide.  usage: python tools/time_wide.py [B] [D ...]     run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import encodermap_amd as em

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Ds = [int(v) for v in sys.argv[2:]] or [1000, 1150, 1300, 2000]
for D in Ds:
    rng = np.random.default_rng(D)
    data = rng.uniform(-np.pi, np.pi, (max(4 * B, 2000), D)).astype(np.float32)
    p = em.Parameters(n_steps=400, batch_size=B, dist_sig_parameters=(45.0, 12, 6, 1, 2, 6))
    e = em.EncoderMap(p, data, read_only=True)
    eng = e._engine
    eng.run_steps(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(200)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    L = eng.losses()
    print(f"B {B} D {D} (network ends {2 * D} wide): {dt * 1e6:8.1f} us / step   loss {L[0]:.4f} -> {L[-1]:.4f}", flush=True)
    del e, eng
