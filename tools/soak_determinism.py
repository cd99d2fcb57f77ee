#!/usr/bin/env python3
"""Soak test of the fused kernel's in-kernel hand-over (round 3: a wave's last chunk slot takes the next wave's first chunk IR
from LDS behind a flag): the same scene rendered many times must give the same bytes every time, for several source
counts (different unit splits, tiles of 8192 and 2048, direct output and slab reduce) - a lost or early flag would show
as a changed sample.   python tools/soak_determinism.py [repeats]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import binaural_audio_synthesis_amd as bas

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
h = bas.synth.make_table("adversarial", 1).truncated(128)
d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
k, s, n = 512, 32, 441000
in_length = -(-n // k) * k
rng = np.random.default_rng(3)
for n_src in (256, 33, 7, 1):
    x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
    x[:, :n] = (torch.rand((n_src, n), device="cuda") * 2 - 1) / n_src
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = torch.from_numpy(rng.uniform(-1.0, 1.7, size=(n_src, t.size))).cuda()
    azim = torch.from_numpy(rng.uniform(-7, 7, size=(n_src, t.size))).cuda()
    ref, _ = bas.apply_hrtf.render_angles_device(x, k, s, d, elev, azim, normalize="none")
    ref = ref.clone()
    bad = 0
    for r in range(reps):
        y, _ = bas.apply_hrtf.render_angles_device(x, k, s, d, elev, azim, normalize="none")
        if not torch.equal(y, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"{n_src:4d} sources: {reps} renders, {bad} differ from the first", flush=True)
    assert bad == 0
print("soak ok")
