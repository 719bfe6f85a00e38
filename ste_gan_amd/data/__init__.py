"""Resident-corpus data path: the corpus lives in HBM, batches are cropped and stacked by one gather kernel."""
from .collate import (UtteranceStore, check_partition, draw_start_frames, load_packed, pack_samples,  # noqa: F401
                      save_packed, ste_gan_collate)
from .loader import DeviceLoader, loaders, loaders_from_samples, loaders_via_config  # noqa: F401
