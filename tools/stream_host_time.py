#!/usr/bin/env python3
"""Host cost of one StreamRenderer.process() call for real-time sized blocks:
256 sources x 512-sample blocks (11.6 ms of audio at 44.1 kHz), audio and trajectories written in place
(input_view / trajectory_views, copy_out=False).  Reports, for graph replay and for plain launches:
  host us/call   time.perf_counter around process() only (no synchronisation): what the caller's thread pays
  wall us/block  many blocks back to back, synchronised at the end: the sustainable rate."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import binaural_audio_synthesis_amd as bas

n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k, s, l, n_blocks = 512, 32, 128, 400
host = bas.synth.make_table("consistent", 0).truncated(l)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
# third argument "two": the round-3 form of a block (render, then the epilogue launch) instead of bas_render_stream_block_f32
if len(sys.argv) > 3 and sys.argv[3] == "two":
    bas.StreamRenderer.one_call = False
form = "one call (carry in the reduce kernel)" if bas.StreamRenderer.one_call else "render + epilogue launch"
for graph in (True, False):
    st = bas.StreamRenderer(tbl, n_src, k, s, graph=graph, copy_out=False)
    xin = st.input_view(B)
    ev, av = st.trajectory_views(B)
    xin.copy_((torch.rand((n_src, B), device="cuda") * 2 - 1) / n_src)
    ev.copy_(torch.rand((n_src, B // k + 1), dtype=torch.float64, device="cuda") - 0.5)
    av.copy_(torch.rand((n_src, B // k + 1), dtype=torch.float64, device="cuda") * 6)
    for _ in range(20):
        st.process(xin, ev, av)
    torch.cuda.synchronize()
    host_us = []
    t_all = time.perf_counter()
    for _ in range(n_blocks):
        t0 = time.perf_counter()
        st.process(xin, ev, av)
        host_us.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / n_blocks * 1e6
    host_us.sort()
    print(f"{n_src} sources x {B}-sample blocks, {form}, {'hipGraph replay' if graph else 'plain launches'}: "
          f"host {np.median(host_us):.1f} us/call (p10 {host_us[len(host_us) // 10]:.1f}, p90 {host_us[9 * len(host_us) // 10]:.1f}), "
          f"wall {wall:.1f} us/block = {B / 44100 * 1e6 / wall:.0f} x real time")
