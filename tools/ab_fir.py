#!/usr/bin/env python3
"""A/B timing of FIR kernel builds in ONE process, interleaved rounds (cdna_hip_programming.md rule 24):
    python tools/ab_fir.py [--sources 256] [--rounds 5] [--reps 20] libA.so libB.so ...
Each library is a build of the same ABI (binaural-audio-synthesis_amd/csrc); per round every library renders the
BASELINE config-4 scene `reps` times through the fused path; the FIR kernel alone is timed with HIP events.
Prints per library: median / min / max of the per-round mean kernel time, and the step time (plans + FIR + reduce)."""
import argparse, ctypes, os, statistics, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import binaural_audio_synthesis_amd as bas
from binaural_audio_synthesis_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--sources", type=int, default=256)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--fused", type=int, default=1)
ap.add_argument("--chunk", type=int, default=512)
ap.add_argument("--subchunk", type=int, default=32)
ap.add_argument("--taps", type=int, default=128)
ap.add_argument("--no-check", action="store_true", help="do not compare the outputs of the builds (diagnostic builds that compute something else on purpose)")
ap.add_argument("--zero-x", action="store_true", help="all-zero input audio: shows how far the clock (power) limits the kernel")
args = ap.parse_args()
n_src, n, k, s, l = args.sources, 441000, args.chunk, args.subchunk, args.taps
host = bas.synth.make_table("consistent", 0).truncated(l)
tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
in_length = -(-n // k) * k
gen = torch.Generator(device="cuda").manual_seed(1)
x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
if not args.zero_x:
    x[:, :n] = (torch.rand((n_src, n), generator=gen, device="cuda") * 2 - 1) / n_src
t = np.arange(0, in_length + 1, k, dtype=np.float64)
elev = np.zeros((n_src, t.size)); azim = np.zeros((n_src, t.size))
for i in range(n_src):
    name = "spiral" if i % 2 == 0 else "circle_askew"
    elev[i], azim[i] = bas.synth.trajectory(name, period_s=2.0 + i / 64.0, length_s=10.0, turns=5.0, phase=2 * np.pi * i / n_src)(t)
idx, w = bas.sphere.interpolation_params_batch(elev, azim)
idx = torch.from_numpy(idx.reshape(-1, 4)).cuda(); w = torch.from_numpy(w.reshape(-1, 3)).cuda()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
def mk():
    e = ctypes.c_void_p(); assert hip.hipEventCreate(ctypes.byref(e)) == 0; return e
y = torch.empty((2, in_length + l - 1), dtype=torch.float32, device="cuda")
ws = bas._hip.new_workspace(1 << 28, "cuda")
wsp = torch.empty((_hip.lib().bas_interp2d_workspace_bytes(idx.shape[0]),), dtype=torch.uint8, device="cuda")
handles = [(_hip.use_library(p if os.path.isabs(p) else os.path.join(os.getcwd(), p)), os.path.basename(p)) for p in args.libs]
res = {name: ([], []) for _, name in handles}
ref = None
worst = {}
tbls = {}                                                      # (the packed table's layout belongs to the build: one per library)
for rnd in range(args.rounds + 1):
    for ctx, name in handles:
        with ctx:
            if name not in tbls:
                tbls[name] = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right)
            tbl = tbls[name]
            evs = [(mk(), mk()) for _ in range(args.reps)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for r in range(args.reps):
                bas.apply_hrtf.render_params_device(x, k, s, tbl, idx, w, normalize="none", out=y, events=evs[r], ws=ws, ws_plans=wsp, fused=bool(args.fused))
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / args.reps * 1e3
            ms = []
            for a, b in evs:
                f = ctypes.c_float(); assert hip.hipEventElapsedTime(ctypes.byref(f), a, b) == 0; ms.append(f.value)
            if ref is None:
                ref = y.clone()
            else:
                scale = float(ref.abs().max())
                err = float((y - ref).abs().max()) / scale if scale > 0 else float(y.abs().max())
                worst[name] = max(worst.get(name, 0.0), err)
                assert args.no_check or err < 1e-5, (name, err)          # (builds may differ in summation order: the tests' tolerance)
            if rnd > 0:
                res[name][0].append(sum(ms) / len(ms)); res[name][1].append(el)
for name, (km, st) in res.items():
    print(f"{name:40s} FIR kernel ms: median {statistics.median(km):.4f} min {min(km):.4f} max {max(km):.4f} | step ms (plans+FIR+reduce) median {statistics.median(st):.4f}"
          + (f" | max rel. difference from the first build {worst[name]:.1e}" if name in worst else ""))
