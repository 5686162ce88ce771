"""Drop-in for the reference's models/video_models/resnet.py."""
from deeplip_amd.video import BasicBlock, ResNet, conv3x3, downsample_basic_block  # noqa: F401
