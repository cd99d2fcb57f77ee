#!/usr/bin/env python3
"""A/B in one process: angles -> parameters -> read plans as two launches (bas_traj_params_branch_f64 + bas_interp2d_plan_f32,
what batches above MERGED_A3_MAX_QUERIES took in round 3) against ONE launch (bas_interp2d_plan_angles_f32: a3 by the first two
waves of a block through LDS, round 4), inside the whole step of the headline scene.   python3 tools/ab_a3_merge.py [sources]"""
import os, statistics, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import apply_hrtf, _hip

n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, k, s, l = 441000, 512, 32, 128
host = bas.synth.make_table("consistent", 0).truncated(l)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
in_length = -(-n // k) * k
x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
x[:, :n] = (torch.rand((n_src, n), device="cuda") * 2 - 1) / n_src
t = np.arange(0, in_length + 1, k, dtype=np.float64)
elev = np.zeros((n_src, t.size)); azim = np.zeros((n_src, t.size))
for i in range(n_src):
    elev[i], azim[i] = bas.synth.trajectory("spiral" if i % 2 == 0 else "circle_askew", period_s=2.0 + i / 64.0, length_s=10.0, turns=5.0, phase=2 * np.pi * i / n_src)(t)
e, a = torch.from_numpy(elev).cuda(), torch.from_numpy(azim).cuda()
lib = _hip.lib()
ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
wsp = torch.empty((lib.bas_interp2d_workspace_bytes(e.numel()),), dtype=torch.uint8, device="cuda")
idx = torch.empty((e.numel(), 4), dtype=torch.int32, device="cuda"); w = torch.empty((e.numel(), 3), dtype=torch.float64, device="cuda")
y = torch.empty((2, in_length + l - 1), dtype=torch.float32, device="cuda")
res, outs = {"two launches": [], "one launch": []}, {}
for rnd in range(7):
    for name, limit in (("two launches", 0), ("one launch", 1 << 30)):
        apply_hrtf.MERGED_A3_MAX_QUERIES = limit
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            apply_hrtf.render_angles_device(x, k, s, tbl, e, a, normalize="mix", out=y, ws=ws, ws_plans=wsp, params=(idx, w))
        torch.cuda.synchronize()
        if rnd:
            res[name].append((time.perf_counter() - t0) / 20 * 1e3)
        outs[name] = y.clone()
assert torch.equal(outs["two launches"], outs["one launch"]), "the merged launch changed the render"
for name, v in res.items():
    print(f"{n_src} sources, {e.numel()} queries, {name:12s}: step median {statistics.median(v):.4f} ms (min {min(v):.4f})")
print("renders bit-identical")
