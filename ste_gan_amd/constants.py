"""Boundary constants of the STE-GAN hot path.

Names and values must equal the reference's ``ste_gan/constants.py`` (star-imported into the
``ste_gan`` namespace, ste_gan/__init__.py:2) because configs, batch dicts and checkpoints are
keyed on them: sample rate / hop (constants.py:11,45), optimiser factory (:57), loss-weight
defaults (:74-106), logging intervals (:113-140), phoneme inventory (:149-151) and the
DataDir / SpeakingMode / DataType key strings (:175-239).
"""
import functools

import torch

# --- signal geometry
EMG_SAMPLE_RATE = 800
SPEECH_UNIT_HOPSIZE_SECONDS = 0.02
HOPSIZE = int(EMG_SAMPLE_RATE * SPEECH_UNIT_HOPSIZE_SECONDS)      # 16 EMG samples per Soft-SU frame
SPEECH_UNITS_FEAT_SIZE = 256
NUM_MFCCS = 25
EMBEDDING_DIM_SIZE = 64
NUM_EMG_CHANNELS = 8
NUM_EMG_SESSIONS = 17
NUM_FEATURES = SPEECH_UNITS_FEAT_SIZE
NUM_DISCRIM_FEATURES = NUM_EMG_CHANNELS

# --- training defaults
BATCH_SIZE = 32
CHUNK_SIZE = 2048
MAX_LENGTH = 10 * EMG_SAMPLE_RATE
MAX_STEPS = 50_000
NUM_WORKERS = 2
RANDOM_SEED = 0
TRAIN_EMG_LENGTH = CHUNK_SIZE
TRAIN_FEATURE_LENGTH = TRAIN_EMG_LENGTH // HOPSIZE
OPTIMIZER = functools.partial(torch.optim.AdamW, lr=2e-4, betas=(.8, .99))

# --- loss switches / weights
LOSS_ADVERSARIAL = 'mse'
LOSS_FEAT_MATCH, LOSS_FEAT_MATCH_WEIGHT = True, 7.
LOSS_MEL_ERROR = True
LOSS_MULTI_TD_ERROR, LOSS_MULTI_TD_ERROR_WEIGHT = True, 15.0
LOSS_SPEECH_UNIT_ERROR, LOSS_SPEECH_UNIT_WEIGHT = True, 1.0
LOSS_PHONEMES_ERROR, LOSS_PHONEMES_WEIGHT = True, 1.0
LOSS_WAVEFORM_ERROR, LOSS_WAVEFORM_ERROR_WEIGHT = False, 1

# --- logging cadence
INTERVAL_LOG = 50
INTERVAL_SAMPLE = 1_000
INTERVAL_EMG_SIGNAL_METRICS = 5_000
INTERVAL_EMG_SYNTH_METRICS = 10_000
INTERVAL_SAVE = 25_000
INTERVAL_VALID = 500
INTERVAL_WAVEFORM = 500
INTERVAL_PLOT = 1000
NUM_TEST_SAMPLES = 10
MIN_NUM_STEPS_ASR_EVAL = 25_000
ASR_EVAL_MORE_TEMPERATURE_MIN_STEPS = 25_000
DEBUG = False

# --- phonemes (ARPAbet, 'sil' last)
PHONEME_INVENTORY = ('aa ae ah ao aw ax axr ay b ch d dh dx eh el em en er ey f g hh hv ih iy jh k l m n nx ng '
                     'ow oy p r s sh t th uh uw v w y z zh sil').split()
SILENCE_PHONEME_INDEX = PHONEME_INVENTORY.index("sil")
NUM_PHONEMS = len(PHONEME_INVENTORY)
EMG_TD_FEAT_NAMES = ["Mean Lowp", "Power Lowp", "Power High", "ZCR High.", "Mean High", "Hilbert Env"]


class DataDir:
    """Sub-directory names of the on-disk corpus."""
    EMG, TRANSCRIPTIONS, PHONEMES, SPEECH_UNITS = "emg", "transcriptions", "phonemes", "units"
    MFCCS, EMG_FEATS, ACOUSTIC_FEATS, AUDIO = "mfccs", "emg_feats", "acoustic_feats", "audio"


class SpeakingMode:
    SILENT, NORMAL = "silent", "normal"


class DataType:
    """Keys of the per-utterance / per-batch dictionaries."""
    UTT_ID = "UTT_ID"
    REAL_EMG = "REAL_EMG"
    EMG_FEATURES = "EMG_FEATS"
    MFCCS = "MFCCS"
    TRANSCRIPTION = "TRANSCRIPTION"
    PHONEMES = "PHONEMES"
    SPEECH_UNITS = "SPEECH_UNITS"
    SESSION_INDEX = "SESSION_INDEX"
    SESSION_ID = "SESSION_ID"
    SPEAKING_MODE_ID = "SPEAKING_MODE"
    SPEAKING_MODE_INDEX = "SPEAKING_MODE_IDX"
    FAKE_EMG = "FAKE_EMG"
    PRED_PHONEMES = "PRED_PHONEMES"
    PRED_SPEECH_UNITS = "PRED_SPEECH_UNITS"
    PRED_TRANSCRIPT = "PRED_TRANSCRIPT"
    PRED_AUDIO_SYNTH = "PRED_SYNTH_AUDIO"
