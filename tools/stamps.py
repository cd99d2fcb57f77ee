#!/usr/bin/env python3
"""Diagnostic: phase shares of the FIR kernel's pass loop (needs `make -C .../csrc stamps`).
Loads libbas_hip_stamps.so in place of the product library, runs the bench workload once and
prints per-phase cycle totals.  Shares only - the stamped build is never timed."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "binaural-audio-synthesis_amd", "csrc", "libbas_hip_stamps.so")
import torch

n_src, n, k, s, l = int(os.environ.get("NSRC", 256)), 441000, 512, 32, 128
host = bas.synth.make_table("consistent", 0).truncated(l)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
x = (torch.rand((n_src, n), device="cuda") - 0.5) / n_src
in_length = -(-n // k) * k
t = np.arange(0, in_length + 1, k, dtype=np.float64)
elev = np.zeros((n_src, t.size)); azim = np.zeros((n_src, t.size))
for i in range(n_src):
    elev[i], azim[i] = bas.synth.trajectory("spiral", length_s=10.0, turns=5.0, phase=i)(t)
for _ in range(3):
    y = bas.render_sources(x, k, s, elev, azim, tbl, normalize="none")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 4 * 8))()
lib = _hip.lib()
lib.bas_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.bas_debug_read_stamps(buf, 1024 * 4 * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(1024, 4, 8)[:512].astype(np.float64)
names = ["issue loads+flush", "FIR", "barrier after FIR", "stage(incl. load wait)", "barrier after stage", "total", "load wait"]
for w in range(4):
    print(f"wave {w}: " + "  ".join(f"{names[i]}={a[:, w, i].mean():.0f}" for i in range(7)))
tot = a[:, :, 5].mean()
print("shares: " + "  ".join(f"{names[i]}={a[:, :, i].mean() / tot:.3f}" for i in (0, 1, 2, 3, 4, 6)))
