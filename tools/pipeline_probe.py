#!/usr/bin/env python3
"""Probe: does running a3 + read plans of step i+1 on a second stream beside the FIR kernel of step i pay?
Prints ms per step for the serial order and for the two-stream pipeline (double-buffered parameters / plans)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip, sphere
from binaural_audio_synthesis_amd.apply_hrtf import render_params_device

n_src, n, k, s, l = 256, 441000, 512, 32, 128
host = bas.synth.make_table("consistent", 0).truncated(l)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
in_length = -(-n // k) * k
x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
x[:, :n] = (torch.rand((n_src, n), device="cuda") * 2 - 1) / n_src
t = np.arange(0, in_length + 1, k, dtype=np.float64)
elev = np.zeros((n_src, t.size)); azim = np.zeros((n_src, t.size))
for i in range(n_src):
    elev[i], azim[i] = bas.synth.trajectory("spiral" if i % 2 == 0 else "circle_askew", period_s=2.0 + i / 64.0, length_s=10.0, turns=5.0, phase=i)(t)
elev = torch.from_numpy(elev).cuda(); azim = torch.from_numpy(azim).cuda()
lib = _hip.lib()
n_q = elev.numel()
ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
y = torch.empty((2, in_length + l - 1), dtype=torch.float32, device="cuda")
bufs = [dict(idx=torch.empty((n_q, 4), dtype=torch.int32, device="cuda"), w=torch.empty((n_q, 3), dtype=torch.float64, device="cuda"),
             plans=torch.empty((lib.bas_interp2d_workspace_bytes(n_q),), dtype=torch.uint8, device="cuda")) for _ in range(2)]
peak = torch.empty(1, dtype=torch.float32, device="cuda")

def prep(b):
    st = _hip.current_stream(x.device)
    sphere.interpolation_params_device(elev, azim, out=(b["idx"], b["w"]))
    _hip.call("bas_interp2d_plan_f32", _hip.ptr(tbl.diffs), _hip.ptr(b["idx"]), _hip.ptr(b["w"]), n_q, tbl.ndir, l, tbl.upsampling,
              _hip.ptr(b["plans"]), b["plans"].numel(), st)

def fir(b):
    st = _hip.current_stream(x.device)
    _hip.call("bas_render_mix_fused_f32", _hip.ptr(x), x.stride(0), _hip.ptr(tbl.packed), _hip.ptr(b["plans"]), n_src, in_length, k, s, l,
              tbl.upsampling, tbl.ndir, _hip.ptr(y), 0, _hip.ptr(peak), 1, _hip.ptr(ws), ws.numel(), st)

steps = 300
for _ in range(100):
    prep(bufs[0]); fir(bufs[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    prep(bufs[0]); fir(bufs[0])
torch.cuda.synchronize(); serial = (time.perf_counter() - t0) / steps * 1e3
ref = y.clone()
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
ev_p = [torch.cuda.Event(), torch.cuda.Event()]
ev_f = [torch.cuda.Event(), torch.cuda.Event()]
def run(nsteps):
    with torch.cuda.stream(side):
        prep(bufs[0]); ev_p[0].record(side)
    for i in range(nsteps):
        b = i & 1
        main.wait_event(ev_p[b])
        fir(bufs[b]); ev_f[b].record(main)
        with torch.cuda.stream(side):                     # plans of step i+1 beside the FIR kernel of step i
            if i >= 1: side.wait_event(ev_f[b ^ 1])       # its buffers were read by the FIR of step i-1
            prep(bufs[b ^ 1]); ev_p[b ^ 1].record(side)
run(50)
torch.cuda.synchronize(); t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize(); piped = (time.perf_counter() - t0) / steps * 1e3
assert torch.equal(ref, y)
print(f"serial {serial:.4f} ms/step, a3 + plans of the next step on a second stream {piped:.4f} ms/step ({100 * (piped / serial - 1):+.1f} %)")
