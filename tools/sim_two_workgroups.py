#!/usr/bin/env python3
"""A two-line model of a CU under the fused FIR kernel: two workgroups share every SIMD (one
wave each); a pass is a staging phase of S microseconds in which the wave issues next to nothing (x window from HBM,
chunk IRs through the vector L1) followed by W microseconds of SIMD time of FIR work.  While both are in their FIR
phases they share the SIMD; while one stages, the other runs alone at `lone_eff` of the SIMD's rate; while both stage,
the SIMD idles.  Event-driven, with jitter on S.  Prints microseconds per pass and CU.

    measured (profiles/r03_stamps_fz_256sources.txt, r03_kernel_stats.csv):  S = 6.6, W = 9.4 at 1.95 GHz -> 11.8 us per pass and CU
    the VALU floor is W = 9.4; what the model says the knobs are worth is below
"""
import random


def sim(S, W=9.4, Sj=1.0, lone_eff=0.89, n_pass=4000, seed=1):
    random.seed(seed)
    wg = [{"ph": "S", "rem": S + random.uniform(-Sj, Sj), "done": 0} for _ in range(2)]
    t = 0.0
    while min(w["done"] for w in wg) < n_pass:
        in_f = [i for i in range(2) if wg[i]["ph"] == "F"]
        rate = [0.0, 0.0]
        if len(in_f) == 2:
            rate = [0.5, 0.5]
        elif len(in_f) == 1:
            rate[in_f[0]] = lone_eff
        dt = min([w["rem"] for w in wg if w["ph"] == "S"] + [wg[i]["rem"] / rate[i] for i in in_f])
        t += dt
        for i, w in enumerate(wg):
            if w["ph"] == "S":
                w["rem"] -= dt
                if w["rem"] <= 1e-12:
                    w["ph"], w["rem"] = "F", W
            else:
                w["rem"] -= dt * rate[i]
                if w["rem"] <= 1e-12:
                    w["done"] += 1
                    w["ph"], w["rem"] = "S", S + random.uniform(-Sj, Sj)
    return t / (wg[0]["done"] + wg[1]["done"])


if __name__ == "__main__":
    for S in (7.5, 6.6, 5.0, 3.0, 0.0):
        for le in (0.8, 0.89, 1.0):
            r = sum(sim(S, lone_eff=le, seed=s) for s in range(4)) / 4
            print(f"staging {S:4.1f} us, lone-wave efficiency {le:4.2f}: {r:5.2f} us per pass and CU")
