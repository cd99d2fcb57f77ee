#!/usr/bin/env python3
"""Diagnostic: where a wave of the fused FIR kernel spends a pass (stamped build, `make -C csrc stamps`).
Phases (10 ns ticks of s_memrealtime, every stamp behind a vmcnt(0)/lgkmcnt(0) wait, so load latency is charged
to the phase that issued the load): plans->LDS | x loads + chunk-IR evaluation | barrier 1 | LDS stores |
barrier 2 | FIR."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip
import torch
n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, k, s, l = 441000, 512, 32, 128
host = bas.synth.make_table("consistent", 0).truncated(l)
with _hip.use_library(os.path.join(ROOT, "binaural-audio-synthesis_amd", "csrc", "libbas_hip_stamps.so")) as lib:
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
    buf = (ctypes.c_ulonglong * (2048 * 4 * 8))()
    lib.bas_debug_read_fz_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.bas_debug_read_fz_stamps(buf, 2048 * 4 * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(2048 * 4, 8).astype(np.float64)
a = a[a[:, 7] > 0]                                           # one row per wave that ran (1 or 4 per workgroup)
print("waves:", a.shape[0], "passes per wave:", a[:, 7].mean())
names = ["issue loads", "barrier 1", "loads->LDS", "eval + slots", "barrier 2", "FIR"]
per_pass = a[:, :6] / a[:, 7:8]
tot = a[:, 6] / a[:, 7]
for i, nm in enumerate(names):
    print(f"{nm:16s} {per_pass[:, i].mean() * 0.01:8.2f} us/pass   (p10 {np.percentile(per_pass[:, i], 10) * 0.01:.2f}, p90 {np.percentile(per_pass[:, i], 90) * 0.01:.2f})")
print(f"{'whole pass':16s} {tot.mean() * 0.01:8.2f} us/pass; kernel ~ {a[:, 6].max() * 0.01:.1f} us")
