#!/usr/bin/env python3
"""Wall time of the drop-in call for BASELINE configs 1-3 (one moving source, 10 s @ 44.1 kHz):
make_signal_move_2d through the HIP path with (a) the reference-style scalar trajectory calls,
(b) vectorized=True, next to the oracle's numpy port on one host core."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from oracle import bas_oracle as orc
import torch

fs, n = 44100, 441000
host = bas.synth.make_table("consistent", 0).truncated(128)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
x = bas.synth.integer_noise(1, n, 0.05)
for name, kw in (("circle_horizontal", {}), ("passing", {}), ("spiral", dict(length_s=10.0, turns=5.0))):
    traj = bas.synth.trajectory(name, fs=fs, **kw)
    for vec in (False, True):
        bas.make_signal_move_2d(x, 512, 32, traj, tbl, vectorized=vec)          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = bas.make_signal_move_2d(x, 512, 32, traj, tbl, vectorized=vec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name:18s} vectorized={vec!s:5s} {dt*1e3:8.2f} ms  = {10.0/dt:9.0f} x real time (host array in, host array out)")
    t0 = time.perf_counter()
    want = orc.render(x, 512, 32, traj, host)
    dt = time.perf_counter() - t0
    err = np.abs(y.astype(np.float64) - want).max() / np.abs(want).max()
    print(f"{name:18s} numpy port, 1 core  {dt*1e3:8.2f} ms  = {10.0/dt:9.1f} x real time   rel err GPU vs port {err:.2e}")
