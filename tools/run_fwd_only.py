import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from encodermap_amd import _lib
from encodermap_amd.layout import build_layout, init_parameters
_lib.load()
B, D = 4096, 128
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
x = torch.randn(B, D, device="cuda")
plan = _lib.MlpPlan(lay, params, h, [None] * n, train_rows=B)
for _ in range(30): _lib.mlp_forward(plan, 0, n, x, 0, B)
torch.cuda.synchronize()
