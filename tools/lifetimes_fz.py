#!/usr/bin/env python3
"""Diagnostic: when the workgroups of the fused FIR kernel start and finish (builds with -DBAS_STAMPS -DBAS_LIFETIME_ONLY:
only a wave's first and last instruction are stamped).   python tools/lifetimes_fz.py lib.so [n_src]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip
import torch
path = os.path.abspath(sys.argv[1])
n_src = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n, k, s, l = 441000, 512, 32, 128
host = bas.synth.make_table("consistent", 0).truncated(l)
with _hip.use_library(path) as lib:
    tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
    x = (torch.rand((n_src, n), device="cuda") - 0.5) / n_src
    in_length = -(-n // k) * k
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.zeros((n_src, t.size)); azim = np.zeros((n_src, t.size))
    for i in range(n_src):
        elev[i], azim[i] = bas.synth.trajectory("spiral", length_s=10.0, turns=5.0, phase=i)(t)
    for _ in range(60):                                       # (the clock governor settles over the first ~40 launches)
        y = bas.render_sources(x, k, s, elev, azim, tbl, normalize="none")
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2048 * 4 * 8))()
    lib.bas_debug_read_fz_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.bas_debug_read_fz_stamps(buf, 2048 * 4 * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.float64)
a = a[a[:, 0, 7] > 0]                                         # workgroups that ran
n_wg = a.shape[0]
begin = a[:, :, 5].min(axis=1); end = (a[:, :, 5] + a[:, :, 6]).max(axis=1)
t0 = begin.min()
begin = (begin - t0) * 0.01; end = (end - t0) * 0.01          # us
h = n_wg // 2
print(f"{os.path.basename(path)}: {n_wg} workgroups, {a[:, 0, 7].mean():.1f} passes each, kernel {end.max():.1f} us")
for name, sl in (("first half of the grid", slice(0, h)), ("second half", slice(h, n_wg))):
    print(f"  {name:24s} start {begin[sl].mean():7.1f} us (max {begin[sl].max():7.1f})   end mean {end[sl].mean():7.1f}  p10 {np.percentile(end[sl], 10):7.1f}  p90 {np.percentile(end[sl], 90):7.1f}  max {end[sl].max():7.1f}")
# end time by XCD (workgroup b runs on XCD b % 8 under round-robin placement) and by position in the grid
print("  end time by b % 8 :", " ".join(f"{end[i::8].mean():6.1f}" for i in range(8)))
print("  end time by b // 64:", " ".join(f"{end[i * 64:(i + 1) * 64].mean():6.1f}" for i in range(n_wg // 64)))
first_src = np.arange(n_wg) * 27 % 256
print("  corr(end, first source of the share) = %.2f" % np.corrcoef(end, first_src)[0, 1])
