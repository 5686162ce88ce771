"""Drop-in for the reference's models/video_models/model.py (same import path and names)."""
from deeplip_amd.video import (Lipreading, MultiscaleMultibranchTCN, TCN, threeD_to_2D_tensor)  # noqa: F401
from deeplip_amd.video import ResNet, BasicBlock  # noqa: F401
