"""The split-f16 arithmetic of the engine against TRUE float32: per quantity of a training step the engine's error against the
float64 oracle must not exceed twice the error of a float32 run of the reference formulation (tests/accuracy.py; the table
itself: tools/split_accuracy.py -> profiles/r03_split_accuracy.txt).  Needs an MI355X."""
import numpy as np
import pytest

from tests import accuracy as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def em():
    import encodermap_amd as em
    from encodermap_amd import _lib

    _lib.require_gpu()
    return em


def _check(table, factor=2.0, floor=2e-7):
    """engine rms error <= factor x float32-oracle rms error (+ a floor of a few float32 ulps: where the float32 oracle happens
    to be exact -- a bias gradient that is a plain sum -- the ratio means nothing)"""
    bad = {k: v for k, v in table.items() if v["engine"][0] > factor * v["oracle_f32"][0] + floor}
    assert not bad, A.format_table("engine worse than 2 x float32:", bad)


@pytest.mark.parametrize("name", list(A.SHAPES))
def test_dense_layers_are_float32_accurate(em, name):
    """Euclidean shapes: 2 x (observed: at most 1.30 x, most quantities 0.3-0.9 x).  The periodic C4 shape: 2.5 x -- behind the atan2
    of its output layer d cost / d (sin, cos) = g (c, -s) / (s^2 + c^2) amplifies the relative error of SMALL outputs, and the
    engine's operands carry an error relative to the largest element of their 32-column block: its decoder-side gradients are
    1.6-1.8 x the float32 run's there (1e-4 both), its activations 0.85 x."""
    table = A.dense_layer_errors(em, name)
    print(A.format_table(name, table))
    _check(table, factor=2.5 if name.startswith("C4") else 2.0)


@pytest.mark.parametrize("name", [n for n in A.SHAPES if not n.startswith("C4")])
def test_weight_gradients_summed_inside_the_workgroups_are_float32_accurate(em, name):
    """the same bound for the weight gradients of emx_mlp_wgrad_adam_local (small batches: no slice partials), read back from Adam's
    first moment after one step"""
    table = {k: v for k, v in A.dense_layer_errors(em, name, local=True).items() if k.startswith("d")}
    print(A.format_table(name + " -- gradients of the one-launch step tail", table))
    _check(table, floor=4e-7)      # (m = 0.1 g and g = 10 m: two more roundings than the stored gradient)


@pytest.mark.parametrize("B,D", [(4096, 128), (1024, 256)])
def test_all_pairs_pass_is_float32_accurate_far_from_the_origin(em, B, D):
    table = A.all_pairs_errors(B, D)
    print(A.format_table(f"all-pairs pass ({B}, {D}), data 40 sigma off the origin", table))
    _check(table)
    # and in absolute terms: the cost to 1e-5, the gradient to 2e-5 of its largest entry (the kernels' own tolerances)
    assert table["distance cost"]["engine"][0] < 1e-5 and table["dZ (latent gradient)"]["engine"][1] < 2e-5
