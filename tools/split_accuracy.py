#!/usr/bin/env python3
"""Writes profiles/r03_split_accuracy.txt: rms and max error against float64 of (engine | float32 run of the reference
formulation) for the dense layers' activations, deltas and weight gradients at the C2 / C4 / C5 widths and for the all-pairs
pass (cost, latent gradient) on data 40 sigma off the origin.  Run on an MI355X:  python tools/split_accuracy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import encodermap_amd as em  # noqa: E402
from tests import accuracy as A  # noqa: E402


def main():
    out = ["Engine (split-f16 x3 matrix products, 22-bit operands, f32 accumulate) and a float32 run of the reference formulation",
           "(oracle/reference_ops.py, torch-CPU float32) against the float64 oracle: error / magnitude of the float64 value, rms over",
           "the tensor and its largest element.  ratio = engine / float32 oracle (< 1: the engine is closer to float64).", ""]
    worst = 0.0
    for name in A.SHAPES:
        t = A.dense_layer_errors(em, name)
        out += [A.format_table(name, t), ""]
        worst = max([worst] + [v["engine"][0] / max(v["oracle_f32"][0], 1e-300) for v in t.values() if v["oracle_f32"][0] > 1e-7])
    for B, D in ((4096, 128), (1024, 256)):
        t = A.all_pairs_errors(B, D)
        out += [A.format_table(f"all-pairs pass B={B} D={D}, data 40 sigma off the origin (Gram products from split rows)", t), ""]
        worst = max([worst] + [v["engine"][0] / max(v["oracle_f32"][0], 1e-300) for v in t.values()])
    out.append(f"largest rms ratio engine / float32 oracle over all quantities: {worst:.2f}")
    text = "\n".join(out)
    print(text)
    dst = os.path.join(ROOT, "gpurun_out", "r03_split_accuracy.txt")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(text + "\n")


if __name__ == "__main__":
    main()
