"""MI355X-native moving-source binaural renderer (hot path of mbjd/binaural-audio-synthesis).

Public surface mirrors the reference's apply_hrtf.py / sphere.py for the path
load_irs_and_delaydiffs -> interpolate_2d -> make_signal_move_2d; all arithmetic
runs in the C-ABI HIP library built from csrc/ (see include/bas.h).
"""
from . import synth, sphere, _hip, apply_hrtf, distributed, stream  # noqa: F401
from .stream import StreamRenderer  # noqa: F401
from .apply_hrtf import (  # noqa: F401
    load_irs_and_delaydiffs, irs_and_delaydiffs, interpolate_2d, interpolate_2d_deg, interpolate_2d_batch,
    interpolate_2d_params, make_signal_move_2d, render_sources, delay_signal_float,
    delay_compensated_interpolation_with_delaydiff, delay_compensated_interpolation,
    delay_compensated_interpolation_easy, make_signal_move)
