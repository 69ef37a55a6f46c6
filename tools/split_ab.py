"""Bit-identity fingerprint of one library build (GPU box): a C2-shaped training run and an inference pass, as sha256 of the
resulting arrays.  tools/run_split_ab.sh runs it once per build (EMX_LIB_PATH) and compares the lines."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import encodermap_amd as em

def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

for periodic, D, B, neurons in ((False, 128, 4096, [128, 128, 2]), (True, 10, 256, [128, 128, 2]), (False, 256, 2048, [256, 128, 2])):
    rng = np.random.default_rng(5)
    data = (rng.uniform(-np.pi, np.pi, (20000, D)) if periodic else rng.normal(0, 3, (20000, D))).astype(np.float32)
    path = f"/tmp/split_ab_{os.getpid()}_{D}"
    os.makedirs(path, exist_ok=True)
    p = em.Parameters(main_path=path, n_steps=24, batch_size=B, n_neurons=neurons,
                      activation_functions=["", "tanh", "tanh", ""], periodicity=2 * np.pi if periodic else float("inf"),
                      dist_sig_parameters=(16.0, 12, 6, 1, 2, 6) if not periodic else (4.5, 12, 6, 1, 2, 6), summary_step=100000,
                      checkpoint_step=100000, seed=7)
    m = em.EncoderMap(p, data)
    h = m.train()
    z = m.encode(data[:5000])
    g = m.generate(z[:3000])
    print(f"periodic={periodic} D={D} B={B}: loss {sha(np.asarray(h.history['loss']))} z {sha(z)} gen {sha(g)} last loss {h.history['loss'][-1]:.8f}", flush=True)
