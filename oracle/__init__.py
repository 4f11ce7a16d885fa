"""CPU oracle for the ColorHandPose3D forward path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the behaviour of lmb-freiburg/hand3d's
hot path (`nets/ColorHandPose3DNetwork.py:61-384`, `nets/PosePriorNetwork.py:59-122`,
`utils/general.py:26-65,112-148,163-357,522-611`) together with the TensorFlow 1.3
kernel semantics those call sites rely on.

Rules (enforced by tests/test_oracle_isolation.py):
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import anything from here;
  * the product package `hand3d_amd` never imports it and has no CPU fallback.

PARITY UNPINNED: the arithmetic of the reference lives in `tensorflow==1.3.0`
(README.md:20-25 of the reference), which is not in /root/reference, cannot be
installed here (no wheel, no network, py3.10) and the reference ships no per-tensor
golden vectors or tests.  The TF op semantics are therefore
restated from the published TF 1.3 kernels and cross-checked in tests/ against
independent implementations (torch-CPU conv/pool/linear, scalar-loop
re-derivations of resize/crop/dilation), not against TF itself.
"""
from . import tf_ops, general, nets  # noqa: F401
