"""Drop-in for the reference's models/audio_models/loss.py."""
from deeplip_amd.loss import AAMSoftmax, ASoftmax, Contrastive, CrossEntropy, LMCL  # noqa: F401
