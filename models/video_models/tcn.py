"""Drop-in for the reference's models/video_models/tcn.py (multibranch MS-TCN)."""
from deeplip_amd.video import (Chomp1d, ConvBatchChompRelu, MultibranchTemporalBlock,  # noqa: F401
                               MultibranchTemporalConvNet)
