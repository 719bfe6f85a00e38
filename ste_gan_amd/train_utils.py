"""Config / CLI helpers of the trainer -- the boundary surface of ``ste_gan/train_utils.py``.

Same function names, flags and "negative / blank = ignore" override convention
(train_utils.py:48-91, 140-190, 204-235); plain host Python, nothing here touches the device.
``omegaconf`` is optional: without it an attribute-access dict with the same read API is used.
"""
from __future__ import annotations

import argparse
import time
from collections import OrderedDict
from typing import Dict

import numpy as np
import yaml

try:                                    # the reference wraps the merged dict in omegaconf.DictConfig
    from omegaconf import DictConfig
except ImportError:                     # not installed in the build image: minimal stand-in
    class DictConfig(dict):
        """Nested attribute-access dict (read API of omegaconf.DictConfig used on this path)."""

        def __init__(self, d=()):
            super().__init__()
            for k, v in dict(d).items():
                self[k] = DictConfig(v) if isinstance(v, dict) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v


# (cli flag, cfg section, cfg key, "is set" predicate)
_OVERRIDES = (
    ("weight_su", "train", "loss_speech_unit_weight", lambda v: v >= 0.0),
    ("weight_phoneme", "train", "loss_phoneme_weight", lambda v: v >= 0.0),
    ("weight_td", "train", "loss_multi_td_weight", lambda v: v >= 0.0),
    ("weight_feat_match", "train", "loss_feat_match_weight", lambda v: v >= 0.0),
    ("speech_feature_type", "model", "speech_feature_type", lambda v: bool(v.strip())),
    ("chunk_size", "train", "chunk_size", lambda v: v > 0),
    ("batch_size", "train", "batch_size", lambda v: v > 0),
    ("max_steps", "train", "max_steps", lambda v: v > 0),
)


def override_config_with_eval_args(cfg: Dict, args) -> Dict:
    """CLI overrides; a loss whose weight ends up < 1e-3 is switched off (train_utils.py:48-91)."""
    print("Overriding config with command-line arguments")
    for flag, section, key, is_set in _OVERRIDES:
        val = getattr(args, flag)
        if is_set(val):
            print(f"cfg.{section}.{key}\t\t{cfg[section][key]} -> {val}")
            cfg[section][key] = val
    for w, sw in (("loss_speech_unit_weight", "loss_speech_unit_error"), ("loss_phoneme_weight", "loss_phoneme_error")):
        if cfg["train"][w] < 0.001:
            cfg["train"][sw] = False
            print(f"Setting cfg.train.{sw} to False")
    return cfg


def mean_error(error_list) -> np.ndarray:
    return np.asarray(error_list).mean(0)


def phoneme_accuracy(num_phones: int, num_correct: int) -> float:
    return 100.0 * (num_correct / num_phones) if num_phones > 0 else float("nan")


def phoneme_accuracy_no_silence(num_phones_total: int, num_correct_no_silence: int, num_silence: int) -> float:
    return phoneme_accuracy(num_phones_total - num_silence, num_correct_no_silence)


def create_ste_gan_model_name(cfg, add_timestamp: bool = True, debug: bool = False, note: str = ""):
    """Output-directory name built from the hyper-parameters (train_utils.py:107-137)."""
    t = cfg.train
    parts = [
        (note + "_") if note else "",
        "DEBUG_" if debug else "",
        "" if (debug or not add_timestamp) else time.strftime("%Y%m%d-%H%M%S") + "_",
        f"{cfg.data.name}_{cfg.model.type}_{cfg.model.speech_feature_type}_",
        "small_dis" if cfg.model.discriminator_small else "full_dis",
        f"_chunk_{t.chunk_size}_",
        "with_adv_loss" if t.loss_adversarial else "no_adv_loss",
        f"_fmw_{t.loss_feat_match_weight}_tdw_{t.loss_multi_td_weight}_suw_{t.loss_speech_unit_weight}"
        f"_phw_{t.loss_phoneme_weight}_wv_{t.loss_waveform_weight}",
    ]
    return "".join(parts)


def add_eval_hyperparams_to_parser(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    for flag, what in (("weight_su", "speech unit loss of the EMG encoder"), ("weight_phoneme", "phoneme loss of the EMG encoder"),
                       ("weight_td", "Multi-Time-Domain loss"), ("weight_feat_match", "feature matching loss")):
        parser.add_argument(f"--{flag}", type=float, default=-1.0, help=f"Weight of the {what} (< 0: ignored)")
    parser.add_argument("--speech_feature_type", type=str, default="",
                        help="DataType of the generator input feature; blank = use the training configuration")
    for flag, what in (("chunk_size", "number of EMG samples per training crop"), ("batch_size", "batch size"),
                       ("max_steps", "maximum number of training steps")):
        parser.add_argument(f"--{flag}", type=int, default=-1, help=f"The {what} (<= 0: ignored)")
    return parser


def fix_state_dict(state_dict):
    """Strip the ``_orig_mod.`` prefix torch.compile'd reference checkpoints carry (train_utils.py:193-201)."""
    return OrderedDict((k.replace("_orig_mod.", ""), v) for k, v in state_dict.items())


def load_config(args: argparse.Namespace, override_with_ste_gan_eval_args: bool = True):
    """Merge model/train, data and encoder YAML into one config (train_utils.py:204-235)."""
    def read(path):
        print(f"Loading {path}")
        with open(path) as fp:
            return yaml.load(fp, Loader=yaml.FullLoader)

    base_cfg = read(args.config)
    data_cfg = read(args.data)
    if args.emg_enc_cfg:
        base_cfg["emg_encoder"] = read(args.emg_enc_cfg)
    base_cfg["data"] = data_cfg
    if override_with_ste_gan_eval_args:
        base_cfg = override_config_with_eval_args(base_cfg, args)
    return DictConfig(base_cfg)
