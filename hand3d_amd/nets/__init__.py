from .ColorHandPose3DNetwork import ColorHandPose3DNetwork  # noqa: F401
from .PosePriorNetwork import PosePriorNetwork  # noqa: F401
