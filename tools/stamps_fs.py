#!/usr/bin/env python3
"""Diagnostic: where the waves of the split-role fused kernel (bas_fused_split.hip) spend their time
(needs `make -C binaural-audio-synthesis_amd/csrc stamps`).  Loads libbas_hip_stamps.so in place of the product
library, renders the bench scene and prints, per role, microseconds per unit of work and of waiting at the
hand-over barrier.  Shares only - the stamped build is never timed."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip
_hip.set_library(os.path.join(ROOT, "binaural-audio-synthesis_amd", "csrc", os.environ.get("STAMPS_LIB", "libbas_hip_stamps.so")))
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
N = 1024 * 8 * 8
buf = (ctypes.c_ulonglong * N)()
lib = _hip.lib()
lib.bas_debug_read_fs_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.bas_debug_read_fs_stamps(buf, N) == 0
a = np.array(buf, dtype=np.uint64).reshape(1024, 8, 8).astype(np.float64)
a = a[a[:, 0, 7] > 0]
units = a[:, :, 7]
print(f"{a.shape[0]} workgroups, {units.mean():.1f} units each, lifetime {a[:, :, 6].mean() / 100:.1f} us")
for w in range(8):
    role = "filter" if w < 4 else "stager"
    print(f"wave {w} ({role}): work {(a[:, w, 0] / units[:, w]).mean() / 100:6.2f} us per unit (p10 {np.percentile(a[:, w, 0] / units[:, w], 10) / 100:.2f}, "
          f"p90 {np.percentile(a[:, w, 0] / units[:, w], 90) / 100:.2f}), barrier wait {(a[:, w, 1] / units[:, w]).mean() / 100:6.2f} us, "
          f"lifetime per unit {(a[:, w, 6] / units[:, w]).mean() / 100:6.2f} us")
ghz = a[:, :, 3].mean() / (a[:, :, 6].mean() * 10.0)
print(f"shader clock over the waves' lifetimes: {ghz:.3f} GHz; filters: {(a[:, :4, 2] / units[:, :4]).mean():.0f} clocks inside the row steps per unit")
