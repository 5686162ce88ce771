"""Drop-in for the reference's models/fusion_models/model_fusion.py."""
from deeplip_amd.fusion import Linearfusion, model_fusion  # noqa: F401
