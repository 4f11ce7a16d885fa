"""CPU oracle for the ColorHandPose3D forward path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the behaviour of lmb-freiburg/hand3d's
hot path (`nets/ColorHandPose3DNetwork.py:61-384`, `nets/PosePriorNetwork.py:59-122`,
`utils/general.py:26-65,112-148,163-357,522-611`) together with the TensorFlow 1.3
kernel semantics those call sites rely on.

Rules (enforced by tests/test_abi.py::test_product_never_imports_oracle_or_frameworks):
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import anything from here;
  * the product package `hand3d_amd` never imports it and has no CPU fallback.

PARITY STATUS.  What the reference itself wrote is pinned by executing it: `oracle/refrun.py`
imports the reference's modules unmodified from /root/reference with `oracle/tfshim` (a NumPy
implementation of the ~70 TensorFlow calls they make) as `tensorflow`, and
tests/test_reference_pin.py holds this package equal to those runs (networks, mask / crop glue,
bone transforms, host helpers, record writer and readers); scripts/make_ref_fixtures.py commits
the same runs as tests/golden/ref_*.npz for the GPU box.
STILL UNPINNED: the arithmetic INSIDE `tensorflow==1.3.0`'s kernels (README.md:20-25 of the
reference).  TF is not in /root/reference, cannot be installed here (no wheel, no network,
py3.10), and the reference ships no per-tensor golden vectors.  tf_ops.py
restates the published TF 1.3 kernel semantics and is cross-checked against independent
implementations (torch-CPU conv / pool / linear, scalar-loop re-derivations of resize / crop /
dilation), not against TF.  scripts/make_tf_fixtures.py is the one-command recipe that closes
this on any box with TF 1.x (-> tests/golden/tf13_*.npz, tests/test_tf13_fixtures.py).
"""
from . import tf_ops, general, nets  # noqa: F401
