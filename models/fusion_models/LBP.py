"""Drop-in for the reference's models/fusion_models/LBP.py."""
from deeplip_amd.fusion import LowFER  # noqa: F401
