"""Checkpoint helpers -- the boundary surface of ``ste_gan/utils/common.py``.

File naming / key layout are the reference's (utils/common.py:23-61, train.py:421-436): ``netG-XXXXXXXX.pt``,
``netD-XXXXXXXX.pt`` hold ``state_dict()``s (possibly with the ``_orig_mod.`` prefix torch.compile adds,
utils/common.py:13-21) and ``checkpoint-XXXXXXXX.pt`` holds ``{epoch, steps, optG, optD}``.
"""
from __future__ import annotations

import logging
from pathlib import Path

import torch
import torch.nn as nn

from ..train_utils import fix_state_dict


def load_latest_checkpoint(checkpoint: Path, device, generator: nn.Module, discriminator: nn.Module, optim_g, optim_d):
    """Restore both nets + optimisers from the highest-numbered ``checkpoint-*.pt`` of a directory."""
    numbered = []
    for f in Path(checkpoint).glob("checkpoint-*.pt"):
        try:
            numbered.append(int(f.stem.split("-")[1]))
        except ValueError:
            pass                                    # checkpoint-final / -last
    if not numbered:
        raise FileNotFoundError(f"no numbered checkpoint-*.pt in {checkpoint}")
    latest = f"{max(numbered):08d}"
    for net, tag in ((generator, "netG"), (discriminator, "netD")):
        path = Path(checkpoint) / f"{tag}-{latest}.pt"
        logging.info(f"LOADING {tag} CHECKPOINT: {path}")
        net.load_state_dict(fix_state_dict(torch.load(path, map_location=device)))
    ckpt = torch.load(Path(checkpoint) / f"checkpoint-{latest}.pt", map_location=device)
    optim_g.load_state_dict(ckpt["optG"])
    optim_d.load_state_dict(ckpt["optD"])
    logging.info(f"START EPOCH {ckpt['epoch']}  STEPS {ckpt['steps']}")
    return generator, discriminator, optim_g, optim_d, ckpt["epoch"], ckpt["steps"]


def initialize_emg_generator(generator: nn.Module, gen_ckpt_file_path: Path, device):
    generator.load_state_dict(fix_state_dict(torch.load(gen_ckpt_file_path, map_location=device)))
    generator.eval()
    return generator
