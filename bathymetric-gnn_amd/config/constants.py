"""Constants shared by training and inference in the reference (``config/constants.py:12,18``)."""

CORRECTION_NORM_FLOOR = 0.01   # metres; floor on local_std when (de)normalising corrections
CORRECTION_NORM_CAP = 50.0     # training-side clamp on normalised corrections (unused at inference)
