"""Golden-fixture case definitions shared by make_golden.py (reference side) and the tests.

Reference-free: only names, shapes and deterministic-input recipes live here.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.detfill import det_input, det_ints  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# ---- generator cases: name -> (feature_type, channels, B, T_feat)
GEN_CASES = {
    "gen_su_c48": ("SPEECH_UNITS", 48, 3, 5),
    "gen_su_c96_t12": ("SPEECH_UNITS", 96, 2, 12),
    "gen_mfcc_c48": ("MFCCS", 48, 2, 6),
    "gen_su_full_c768": ("SPEECH_UNITS", 768, 1, 3),
    "gen_su_c48_t1": ("SPEECH_UNITS", 48, 1, 1),         # single-frame edge case (T_emg=16)
}

# ---- discriminator cases: name -> (small, B, T)
DISC_CASES = {
    "disc_small_t192": (True, 2, 192),
    "disc_small_t64": (True, 3, 64),
    "disc_small_t400": (True, 1, 400),
    "disc_full_t192": (False, 1, 192),
}

# ---- TD-loss cases: name -> (B, T)
TD_CASES = {"td_t192": (2, 192), "td_t400": (1, 400), "td_t96": (3, 96)}

# ---- encoder cases: name -> (model_size, n_layers, dim_ff, B, T_emg)
ENC_CASES = {
    "enc_m64_l2": (64, 2, 128, 2, 192),
    "enc_m64_l2_long": (64, 2, 128, 1, 1664),             # T_enc=104 > 100 -> -1e8 mask branch
    "enc_full_t64": (768, 6, 3072, 1, 64),
}

# ---- train-step cases: name -> (gen channels, enc (model_size, layers, ff), B, T_su, steps)
# (gradients are discontinuous at ReLU / leaky-ReLU kinks: make_golden.py asserts that no pre-activation of the
# chosen case lies within 1e-6 of zero (fp32 path differences are ~1e-7), so fp32 rounding differences cannot flip a mask -- the first candidate,
# B=2 T_su=4, had one generator activation at 1.6e-7 and was rejected for that reason)
STEP_CASES = {"step_c48_b3": (48, (64, 2, 128), 3, 5, 2)}

# ---- batch-assembly cases (data/collate.py): name -> (frames per utterance, extra EMG samples past 16*frames,
#      emg_train_length, n_mfcc, torch seed).  chunk 200 is not a multiple of the hop: the reference crops 12 frames but
#      200 samples, which only stacks when the drawn window fits, hence the 8 extra samples (a non-strict corpus).
COLLATE_CASES = {
    "collate_c200": ((40, 13, 97, 12, 64, 30, 21), 8, 200, 25, 1234),
    "collate_c2048": ((130, 128, 400, 251), 0, 2048, 25, 7),             # the reference default chunk
}


# ---- TimeDomainFeatureLoss with non-default options: name -> (B, T, win, shift, apply_padding_windowing, filter width);
#      td_long: the default multi-scale loss on a row too long for LDS (whole-utterance validation)
TDOPT_CASES = {"tdopt_nopad_w5": (2, 160, 21, 8, False, 5), "tdopt_pad_w3": (1, 96, 30, 7, True, 3),
               "tdopt_nopad_w9": (2, 200, 51, 13, False, 9)}
TD_LONG = ("td_long_t6000", 1, 6000)
# ---- AverageFilter: name -> (B, C, T, window, pad_signal)
AVG_CASES = {"avg_w9_pad": (2, 8, 50, 9, True), "avg_w5_nopad": (2, 3, 40, 5, False), "avg_w9_short": (1, 8, 6, 9, True)}

# ---- validation pass (train.py:292-378): name -> (gen channels, enc (model_size, layers, ff), [(B, T_su), ...])
VALID_CASES = {"valid_c48": (48, (64, 2, 128), [(2, 32), (2, 20), (1, 300)])}

# ---- dataset selection (data/emg_dataset.py:50-151,220-262): partition -> [(utterance id, frames, extra EMG samples)];
#      ids are {session}__{utterance}__{speaking mode}
CORPUS = {
    "train": [("s3__u01__normal", 40, 0), ("s1__u02__normal", 12, 0), ("s2__u03__silent", 50, 0), ("s1__u04__normal", 13, 0),
              ("s2__u05__normal", 20, 8), ("s10__u06__normal", 33, 0), ("s1__u07__whisper", 25, 0)],
    "valid": [("s2__v01__normal", 30, 0), ("s1__v02__silent", 30, 0), ("s3__v03__normal", 5, 0)],
    "test": [("s1__t01__normal", 6, 0), ("s10__t02__normal", 44, 0), ("s2__t03__silent", 9, 0)],
}
CORPUS_EMG_LENGTH = 200


def write_corpus(root, strict_ok: bool = True):
    """The synthetic on-disk corpus in the reference's layout (<root>/<partition>/{emg,units,phonemes,mfccs,emg_feats,
    transcriptions}/<id>.{pt,txt}); deterministic content.  ``strict_ok=False`` adds the 8 extra samples of "u05"."""
    import torch
    from pathlib import Path
    root = Path(root)
    for part, lst in CORPUS.items():
        for sub in ("emg", "units", "phonemes", "mfccs", "emg_feats", "transcriptions"):
            (root / part / sub).mkdir(parents=True, exist_ok=True)
        for uid, f, extra in lst:
            extra = 0 if strict_ok else extra
            torch.save(det_input(uid + ".emg", (16 * f + extra, 8), "tanh"), root / part / "emg" / f"{uid}.pt")
            torch.save(det_input(uid + ".su", (f, 256)), root / part / "units" / f"{uid}.pt")
            torch.save(det_ints(uid + ".ph", (f,), 48), root / part / "phonemes" / f"{uid}.pt")
            torch.save(det_input(uid + ".mf", (2 * f, 25)), root / part / "mfccs" / f"{uid}.pt")
            torch.save(det_input(uid + ".ef", (f, 112)), root / part / "emg_feats" / f"{uid}.pt")
            (root / part / "transcriptions" / f"{uid}.txt").write_text(f"Utterance {uid}\n")


def valid_batches(name):
    ch, enc, shapes = VALID_CASES[name]
    out = []
    for i, (B, T) in enumerate(shapes):
        tag = f"{name}.b{i}"
        su = det_input(tag + ".su", (B, T, 256))
        out.append({"su": su, "feats": su, "x_real": det_input(tag + ".x", (B, 16 * T, 8), "tanh"),
                    "sess": det_ints(tag + ".sess", (B,), 17), "phonemes": det_ints(tag + ".ph", (B, T), 48)})
    return out


def poscheck(t):
    """Checksum that moves when any element changes or shifts: float64 [sum, sum(x*w), sum(x*x)], w = 1 + (i mod 8191)/8191."""
    import numpy as np
    x = np.asarray(t, dtype=np.float64).reshape(-1)
    w = 1.0 + (np.arange(x.size) % 8191) / 8191.0
    return np.array([x.sum(), (x * w).sum(), (x * x).sum()])


def collate_samples(name):
    """Synthetic utterances shaped like EMGDataset.__getitem__ output (data/emg_dataset.py:277-294); keys are the
    DataType strings (constants.py:192-239)."""
    frames, extra, _, n_mfcc, _ = COLLATE_CASES[name]
    out = []
    for i, f in enumerate(frames):
        tag = f"{name}.u{i}"
        out.append({"REAL_EMG": det_input(tag + ".emg", (16 * f + extra, 8), "tanh"),
                    "SPEECH_UNITS": det_input(tag + ".su", (f, 256)),
                    "PHONEMES": det_ints(tag + ".ph", (f,), 48),
                    "SESSION_INDEX": det_ints(tag + ".sess", (1,), 17)[0],
                    "SPEAKING_MODE_IDX": det_ints(tag + ".spk", (1,), 3)[0],
                    "MFCCS": det_input(tag + ".mfcc", (2 * f, n_mfcc))})
    return out


def gen_inputs(name):
    ft, ch, B, T = GEN_CASES[name]
    F = 256 if ft == "SPEECH_UNITS" else 25
    return (det_input(name + ".feats", (B, T, F)), det_ints(name + ".sess", (B,), 17),
            det_ints(name + ".spk", (B,), 3))


def disc_input(name):
    small, B, T = DISC_CASES[name]
    return det_input(name + ".x", (B, T, 8), "tanh")


def td_inputs(name):
    B, T = TD_CASES[name]
    return det_input(name + ".real", (B, T, 8), "tanh"), det_input(name + ".gen", (B, T, 8), "tanh")


def enc_inputs(name):
    ms, nl, ff, B, T = ENC_CASES[name]
    Te = T // 16
    return (det_input(name + ".emg", (B, T, 8), "tanh"), det_input(name + ".su", (B, Te, 256)),
            det_ints(name + ".ph", (B, Te), 48))


def step_batch(name, step):
    ch, enc, B, T, _ = STEP_CASES[name]
    tag = f"{name}.s{step}"
    su = det_input(tag + ".su", (B, T, 256))
    return {"su": su, "feats": su, "x_real": det_input(tag + ".x", (B, 16 * T, 8), "tanh"),
            "sess": det_ints(tag + ".sess", (B,), 17), "phonemes": det_ints(tag + ".ph", (B, T), 48)}
