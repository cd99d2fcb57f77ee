#!/usr/bin/env python3
"""bas_mix_finish_f32 (the root's fixed-order sum + max|y| + peak rule, one launch) by number of parts: what the root of an
N-GPU group pays per step on top of its own render, beside the gather itself.  One GPU, HIP-event timing of 200 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip

t_out = 441344 + 127
lib = _hip.lib()
ws = _hip.new_workspace(lib.bas_mix_workspace_bytes(), "cuda")
y = torch.empty((2, t_out), dtype=torch.float32, device="cuda")
peak = torch.empty((1,), dtype=torch.float32, device="cuda")
for n_parts in (1, 2, 4, 8):
    parts = (torch.rand((n_parts, 2, t_out), device="cuda") - 0.5) / n_parts
    def go():
        _hip.call("bas_mix_finish_f32", _hip.ptr(parts), n_parts, 2 * t_out, 2 * t_out, _hip.ptr(y), _hip.ptr(peak), 1,
                  _hip.ptr(ws), ws.numel(), _hip.current_stream(parts.device))
    for _ in range(20):
        go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(200):
        go()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 200 * 1e3
    mb = (n_parts + 1) * 2 * t_out * 4 / 1e6
    print(f"{n_parts} parts of {2 * t_out * 4 / 1e6:.1f} MB: {us:6.1f} us per launch back to back ({mb:.1f} MB moved: {mb / us:.2f} TB/s)")
