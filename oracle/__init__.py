"""CPU oracle for the EncoderMap training hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``encodermap_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker, never as the thing measured or shipped.

What it is
----------
A CPU restatement of the arithmetic the reference (AG-Peter/encodermap) runs on
TensorFlow for one training step.  Two
independent statements are kept so they can check each other:

``oracle.reference_ops``
    dtype-generic PyTorch-CPU code that follows the reference *formulation* op
    by op (Gram-matrix ``pairwise_dist`` with its ``d==0`` mask, ``[B,B,D]``
    periodic temporaries, separate sigmoid passes, autograd for the backward).
    Run in float64 it is the oracle; run in float32 it is the timed
    "reference-formulation CPU path" (``cpu_baseline.kind == "port"``).
``oracle.direct_ops``
    NumPy float64 direct-difference formulas with hand-derived analytic
    gradients -- the same math the HIP kernels implement.

Pinning
-------
The reference cannot be imported in the build container (TensorFlow and a dozen
other import-time dependencies are absent), so no golden
vector could be captured from the reference's own execution.  The oracle is
pinned by the known-answer values the reference's own tests assert which are
reproducible with numpy/scipy alone (``tests/test_oracle_kat.py`` lists each one
with its reference file:line).  For everything that only TensorFlow could pin
(loss trajectory over steps, Adam/clipvalue numerics, initializer streams,
shuffle order) the status is **parity unpinned**: the claim is "matches this
float64 oracle loop on identical injected weights and batch indices".
"""
