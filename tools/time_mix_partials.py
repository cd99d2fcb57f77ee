#!/usr/bin/env python3
"""Time of rank 0's fixed-order sum of the gathered partial mixes (bas_mix_partials_f32) for 2, 4 and 8 ranks' worth of
[2][T_out] parts on one GPU:   python tools/time_mix_partials.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import distributed as D

t_out = 441344 + 127
for p in (2, 4, 8):
    parts = torch.rand((p, 2, t_out), device="cuda") - 0.5
    y, peak = D._hip_mix_partials(parts)
    ref = parts.sum(0)
    assert float((y - ref).abs().max()) < 1e-5 and abs(float(peak) - float(ref.abs().max())) < 1e-5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(200):
        D._hip_mix_partials(parts)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{p} parts x [2][{t_out}]: {ev[0].elapsed_time(ev[1]) / 200 * 1e3:.1f} us per call (incl. the peak memset and output allocation)")
