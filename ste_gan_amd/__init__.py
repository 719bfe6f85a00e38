"""ste_gan_amd -- MI355X-native STE-GAN hot path behind the reference's Python API.

Drop-in for the ``ste_gan`` package on that path: same constants namespace, module factories
(``init_emg_generator`` / ``init_emg_discriminators`` / ``load_emg_encoder``), loss modules,
state-dict layout and ``train`` entry point; all device arithmetic runs in libstegan_hip.so.
"""
from .constants import *  # noqa: F401,F403
