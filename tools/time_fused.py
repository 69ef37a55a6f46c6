"""The fused dense-layer launch (forward + auto cost + decoder backward) alone, inside a hipGraph of 50 launches, over batch sizes --
the row-block chain kernel against the transposed chain (EMX_TCHAIN is read once per process: one process per variant).
usage: [EMX_TCHAIN=0|1] python tools/time_fused.py [D] [hidden] [B ...]      run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters

L.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Bs = [int(v) for v in sys.argv[3:]] or [256, 1024, 4096, 8192, 16384, 65536]
n_neurons = [H, H, 2]
afs = [""] + ["tanh"] * (len(n_neurons) - 1) + [""]
lay = build_layout(n_neurons, afs, D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 9)).cuda()
for B in Bs:
    xb = ((torch.rand(B, D) - 0.5) * 6.0).cuda()
    h = [torch.zeros(B, l.n_out, device="cuda") if i < n - 1 else None for i, l in enumerate(lay.layers)]
    dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
    dlat = torch.zeros(B, lay.layers[lay.n_enc - 1].n_out, device="cuda")
    plan = L.MlpPlan(lay, params, h, dl, train_rows=B)
    cost_ws = torch.zeros(L.mlp_row_blocks(B), device="cuda")
    cost = L.make_auto_cost(xb, D, 0, D, False, float("inf"), "mean_abs", 0.7, B, dl[-1], cost_ws)
    for _ in range(3):
        L.mlp_forward_backward(plan, 0, n, xb, 0, B, cost, lay.n_enc, dbelow=dlat)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(50):
                L.mlp_forward_backward(plan, 0, n, xb, 0, B, cost, lay.n_enc, dbelow=dlat)
    best = 1e9
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
    print(f"EMX_TCHAIN={os.environ.get('EMX_TCHAIN', '')} D {D} hidden {H} B {B:6d}: {best:8.2f} us per fused launch  cost {float(cost_ws.sum()):.6f}", flush=True)
