"""Drop-in for the reference's models/audio_models/tdnn.py."""
from deeplip_amd.audio import SpeakerEmbNet, TDNN_Block  # noqa: F401
from deeplip_amd.audio import MeanStdPooling, AttentiveStatPooling  # noqa: F401  (tdnn.py:5 star-imports pooling)
