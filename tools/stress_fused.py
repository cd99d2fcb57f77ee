#!/usr/bin/env python3
"""Randomised check of the fused path (every kernel behind bas_render_mix_fused_f32: split roles on tiles of 8192 with and
without the unit block, four waves per tile of 2048, one wave per tile, two workgroups per CU with (h0, d) or h-only rows;
direct output, slab reduce, wide reduce, several tap segments) against the oracle on the adversarial table: random IR lengths, chunk / subchunk sizes,
source counts and signal lengths, random (not smooth) trajectories.   python tools/stress_fused.py [cases] [seed] [split]
"split": through the diagnostic build with BAS_FZ_SPLIT=1, which gives every scene with at least one (tile of 8192, source) unit
per CU the split-role kernel (bas_fused_split.hip; the shipped library asks for more than one)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binaural_audio_synthesis_amd as bas
from oracle import bas_oracle as orc

force_split = len(sys.argv) > 3 and sys.argv[3] == "split"
if force_split:
    os.environ["BAS_FZ_SPLIT"] = "1"
    bas._hip.set_library(bas._hip.DIAG_LIB_PATH)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
full = bas.synth.make_table("adversarial", 1)
lib = bas._hip.lib()
worst, seen = 0.0, {}
t_start = time.time()
for case in range(cases):
    l = int(rng.choice([1, 7, 64, 100, 128, 128, 128, 129, 200, 300]))
    if force_split and case % 2 == 0:
        l = 128 - case % 8                                    # every other case: one whole 128-tap segment (L = 122 .. 128)
    s = 32 * int(rng.integers(1, 9))
    small_k = rng.random() < (0.1 if force_split else 0.3)                            # chunk sizes 256 .. 447: the fused kernel's h-only rows
    if small_k:
        s = 32 * int(rng.integers(1, 5))
        k = s * int(rng.integers(-(-256 // s), max(-(-256 // s) + 1, 447 // s + 1)))
    else:
        k = s * int(rng.integers(max(1, -(-448 // s)), max(2, 4096 // s) + 1))
    sub = 0
    if not small_k and rng.random() < 0.3:                     # subchunks of 16 / 8: the unit blocks with two / four tap sets per row
        sub = int(rng.choice([16, 8]))
        s = sub
        k = 32 * int(rng.integers(14, 65))                     # (the unit blocks exist for K >= 448)
        l = int(rng.choice([128, 125, 100, 104, 97, 121]))
    big = small_k or sub or rng.random() < (0.7 if force_split else 0.35)              # enough (tile, source) units for tiles of 8192
    n_src = int(rng.integers(24, 48)) if big else int(rng.integers(1, 9))
    n = int(rng.integers(100000, 160000)) if big else int(rng.integers(1, 40000))
    h = full.truncated(l)
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    in_length, _ = orc.render_lengths(n, k, l)
    if not lib.bas_render_fused_supported(n_src, in_length, k, s, l):
        continue
    sigs = np.stack([bas.synth.integer_noise(int(rng.integers(1e6)), n, 0.5 / n_src) for _ in range(n_src)])
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = rng.uniform(-1.0, 1.7, size=(n_src, t.size)); azim = rng.uniform(-7, 7, size=(n_src, t.size))
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(t.size)]) for i in range(n_src)]
    want = orc.render_mix(sigs, k, s, irs, normalize=False)
    got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=True).cpu().numpy()
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    worst = max(worst, err)
    kind = lib.bas_render_fused_kernel_name(n_src, in_length, k, s, l).decode().replace("bas_render_", "")   # the kernel the plan picked
    seen[kind] = seen.get(kind, 0) + 1
    print(f"case {case:3d} L={l:4d} K={k:5d} S={s:4d} n_src={n_src:3d} n={n:6d} {kind} rel err {err:.2e}", flush=True)
    assert got.shape == want.shape and err < 1e-5, "PARITY FAILURE"
print("worst", worst, seen, f"{time.time() - t_start:.0f} s")
