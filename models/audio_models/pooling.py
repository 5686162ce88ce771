"""Drop-in for the reference's models/audio_models/pooling.py."""
from deeplip_amd.audio import AttentiveStatPooling, MeanStdPooling  # noqa: F401
