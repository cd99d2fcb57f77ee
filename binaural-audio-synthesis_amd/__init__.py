"""MI355X-native moving-source binaural renderer (hot path of mbjd/binaural-audio-synthesis).

Public surface mirrors the reference's apply_hrtf.py / sphere.py for the path
load_irs_and_delaydiffs -> interpolate_2d -> make_signal_move_2d; all arithmetic
runs in the C-ABI HIP library built from csrc/ (see include/bas.h).
"""
from . import synth  # noqa: F401
