"""hand3d_amd -- MI355X (gfx950) inference engine behind the call surface of
lmb-freiburg/hand3d's ColorHandPose3DNetwork / PosePriorNetwork.

Python host code (this package) + libhp3d.so (hand-written HIP, C ABI in include/hp3d.h).
"""
from .nets import ColorHandPose3DNetwork, PosePriorNetwork  # noqa: F401
from ._lib import Engine, Hp3dError  # noqa: F401
