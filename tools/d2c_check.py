"""Accuracy of the back-mapping kernels against the float64 oracle, with the float32 run of the oracle (the reference formulation
in the reference's precision) as the yardstick: forward coordinates, bond lengths, and both gradients."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from encodermap_amd import _lib as L
from oracle import reference_ops as R
from tests.test_backmap_gpu import make_chain, dev, PI
def run_oracle(dt, ml, angles, dih, gout):
    a = torch.tensor(angles, dtype=dt, requires_grad=True); d = torch.tensor(dih, dtype=dt, requires_grad=True)
    plane = R.chain_in_plane(torch.tensor(ml, dtype=dt)[None, :], a)
    xyz = R.dihedrals_to_cartesian(d + PI, plane)
    (xyz * torch.tensor(gout, dtype=dt)).sum().backward()
    return xyz.detach().double().numpy(), d.grad.double().numpy(), a.grad.double().numpy()
for B, A in [(40, 60), (64, 228), (8, 457)]:
    rng = np.random.default_rng(2)
    lengths, angles, dih = make_chain(rng, B, A)
    ml = lengths.astype(np.float64).mean(0).astype(np.float32)
    gout = rng.normal(0, 1, (B, A, 3)).astype(np.float32)
    x64, gd64, ga64 = run_oracle(torch.float64, ml, angles, dih, gout)
    x32, gd32, ga32 = run_oracle(torch.float32, ml, angles, dih, gout)
    plane, out = torch.empty(B, A, 3, device="cuda"), torch.empty(B, A, 3, device="cuda")
    L.chain_in_plane_fwd(dev(ml), dev(angles), plane)
    L.dihedrals_to_cartesian_fwd(dev(dih), PI, plane, out)
    dd, dp, da = torch.empty(B, A - 3, device="cuda"), torch.empty(B, A, 3, device="cuda"), torch.empty(B, A - 2, device="cuda")
    L.dihedrals_to_cartesian_bwd(dev(dih), PI, plane, dev(gout), dd, dp)
    L.chain_in_plane_bwd(dev(ml), dev(angles), dp, da)
    x = out.cpu().numpy().astype(np.float64)
    bl = lambda v: np.linalg.norm(np.diff(v, axis=1), axis=2)
    rms = lambda v: float(np.sqrt(np.mean(v ** 2)))
    print(f"B={B} A={A} extent {np.abs(x64).max():.0f}:  rms error  kernel / oracle-f32   xyz {rms(x - x64):.2e} / {rms(x32 - x64):.2e}   bonds {rms(bl(x) - bl(x64)):.2e} / {rms(bl(x32) - bl(x64)):.2e}"
          f"   d dihedrals {rms(dd.cpu().numpy() - gd64) / rms(gd64):.2e} / {rms(gd32 - gd64) / rms(gd64):.2e}   d angles {rms(da.cpu().numpy() - ga64) / rms(ga64):.2e} / {rms(ga32 - ga64) / rms(ga64):.2e}")
