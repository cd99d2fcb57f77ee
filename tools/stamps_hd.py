#!/usr/bin/env python3
"""Diagnostic: per-workgroup start/end times and placement of the hd FIR kernel (stamped build)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, "binaural-audio-synthesis_amd", "csrc", "libbas_hip_stamps.so")
import torch
n_src, n, k, s, l = 256, 441000, 512, 32, 128
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
a = np.array(buf, dtype=np.uint64).reshape(1024, 4, 8)[:, 0, :]
a = a[a[:, 0] > 0]
print('workgroups:', a.shape[0])
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
base = t0.min()
st = (t0 - base) / 100.0; en = (t1 - base) / 100.0      # microseconds (100 MHz counter)
print("kernel span us:", en.max(), " start spread us: p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [50, 90, 100])))
print("end times us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(en, [10, 50, 90, 100])))
print("durations us: min %.1f p50 %.1f max %.1f" % tuple(np.percentile(en - st, [0, 50, 100])))
hw = a[:, 3]; xcc = a[:, 4] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 10 * 2 + cu
import collections
cnt = collections.Counter(key.tolist())
print("distinct (xcc,se,sh,cu):", len(cnt), "WGs per CU histogram:", collections.Counter(cnt.values()))
late = st > 50
print("WGs starting later than 50us:", int(late.sum()))
print("fir share of workgroup lifetime: %.3f" % (a[:, 2].astype(np.float64).sum() / (t1 - t0).sum()))
order = np.argsort(en)
print("fir share, earliest-finishing quarter: %.3f, latest quarter: %.3f" % (
    a[order[:len(order)//4], 2].astype(np.float64).sum() / (t1 - t0)[order[:len(order)//4]].sum(),
    a[order[-len(order)//4:], 2].astype(np.float64).sum() / (t1 - t0)[order[-len(order)//4:]].sum()))

# per-CU pairs and per-XCC means
pairs = collections.defaultdict(list)
for i in range(a.shape[0]):
    pairs[int(key[i])].append((float(en[i]), int(xcc[i]), i))
first = np.array([min(v)[0] for v in pairs.values() if len(v) == 2])
second = np.array([max(v)[0] for v in pairs.values() if len(v) == 2])
print("per CU: first-finisher end us mean %.1f (min %.1f max %.1f); second-finisher mean %.1f (min %.1f max %.1f)" % (
    first.mean(), first.min(), first.max(), second.mean(), second.min(), second.max()))
for xc in range(8):
    sel = xcc == xc
    print("xcc", xc, "n", int(sel.sum()), "end mean %.1f" % en[sel].mean(), "min %.1f max %.1f" % (en[sel].min(), en[sel].max()))
# is the first finisher the lower block index (dispatched first)?
lower_first = sum(1 for v in pairs.values() if len(v) == 2 and min(v)[2] == min(x[2] for x in v))
print("CUs where the lower blockIdx finished first:", lower_first, "of", len(pairs))
