"""Drop-in for the module the reference imports as ``models.resnet`` (train_audio.py:65) for ``arch: resnet``; no
source ships upstream, so the architecture is build-owned (deeplip_amd/audio_resnet.py)."""
from deeplip_amd.audio_resnet import SpeakerEmbNet  # noqa: F401
