#!/usr/bin/env python3
"""Interpreter for the unit blocks tools/gen_fir_asm.py writes (one lane, float64): executes the generated instruction list
- ds_read_b128 with the in-order LDS return queue and s_waitcnt lgkmcnt(N), v_pk_fma_f32 / v_pk_add_f32 with op_sel /
op_sel_hi, v_add_f32, v_fma_f32, v_mov_b32 - on random x rows and taps, combines the accumulators the way the kernels'
flush does and compares with the direct sum  y[o] = sum_r sum_a x[R-r][a] * g_r[o - a + 32 r]  (bas_fir.h; apply_hrtf.py:442-446).
It also checks the SCHEDULE: a register is never read while an LDS read into it is still in flight (missing wait), and never
overwritten by a read that was issued before its last use (a read issued too early).  CPU only: tests/test_round4_cpu.py."""
import re

import numpy as np

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def parse_reg(tok):
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), 1
    return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1


def sel_bits(line, name, default):
    m = re.search(name + r":\[(\d),(\d),(\d)\]", line)
    return [int(m.group(i)) for i in (1, 2, 3)] if m else list(default)


def run(lines, xr_stride, lseg, nsub=1, psplit=False, seed=0):
    """Returns (y_block [32, 2], y_direct [32, 2])."""
    rng = np.random.default_rng(seed)
    halo = (lseg + 31) >> 5
    n_steps = halo + 1
    x = rng.standard_normal((n_steps, 32))                     # x[r] = the input row the lane meets in step r (R - r)
    h0 = rng.standard_normal((n_steps, 64, 2))                 # tap t of step r: (h0_L, h0_R), (d_L, d_R); tap k = 32 r - 32 + t
    d = rng.standard_normal((n_steps, 64, 2))
    al = rng.uniform(0, 1, size=n_steps)
    dl = 1.0 / 64.0
    bl = al + dl
    v = np.full(256, np.nan)
    v[:116 if psplit else 98] = 0.0                            # accumulators start at zero
    pending = []                                               # in-flight LDS reads: (first reg, values, issue index)
    last_read_at = np.full(256, -1)                            # instruction index of a register's last read
    operands = {f"al{r}": al[r] for r in range(n_steps)}
    operands.update({f"bl{r}": bl[r] for r in range(n_steps)})
    operands["dl"] = dl

    def read(reg, count, at):
        for k in range(reg, reg + count):
            for p0, vals, _ in pending:
                assert not (p0 <= k < p0 + len(vals)), f"instruction {at}: v{k} read while an LDS read into it is in flight"
            assert not np.isnan(v[k]), f"instruction {at}: v{k} read before it was written"
            last_read_at[k] = at
        return v[reg:reg + count].copy()

    for at, ln in enumerate(lines):
        op = ln.split()[0]
        if op == ".p2align":
            continue
        if op == "s_waitcnt":
            n = int(re.search(r"lgkmcnt\((\d+)\)", ln).group(1))
            while len(pending) > n:
                p0, vals, issued = pending.pop(0)
                v[p0:p0 + len(vals)] = vals
            continue
        toks = [t for t in re.split(r"[ ,]+", ln) if t]
        if op == "ds_read_b128":
            reg, cnt = parse_reg(toks[1])
            off = int(re.search(r"offset:(\d+)", ln).group(1))
            for k in range(reg, reg + cnt):                    # the registers must not be needed any more by earlier instructions:
                pass                                           # (checked when they ARE read later: a pending entry blocks reads)
            if "%[xrow]" in ln:
                c, rem = divmod(off, xr_stride * 16)
                r = halo - rem // 16
                vals = x[r, 4 * c:4 * c + 4].copy()
            else:
                r = int(re.search(r"%\[tap(\d)\]", ln).group(1))
                t = off // 16
                vals = np.array([h0[r, t, 0], h0[r, t, 1], d[r, t, 0], d[r, t, 1]])
            pending.append((reg, vals, at))
            v[reg:reg + cnt] = np.nan                          # old contents are gone as far as later readers are concerned
            continue
        dst, dcnt = parse_reg(toks[1])
        srcs = []
        for tok in toks[2:]:
            if tok.startswith("op_sel"):
                break
            pr_ = parse_reg(tok)
            if pr_ is not None:
                srcs.append(("v",) + pr_)
            elif tok.startswith("%["):
                srcs.append(("c", operands[tok[2:-1]]))
            else:
                srcs.append(("c", float(tok)))
        if op in ("v_pk_fma_f32", "v_pk_add_f32"):
            lo = sel_bits(ln, "op_sel", (0, 0, 0))
            hi = sel_bits(ln, "op_sel_hi", (1, 1, 1))
            out = []
            for half, sel in ((0, lo), (1, hi)):                # (only the halves op_sel picks are read)
                a = [read(srcs[i][1] + sel[i], 1, at)[0] for i in range(len(srcs))]
                out.append(a[0] * a[1] + a[2] if op == "v_pk_fma_f32" else a[0] + a[1])
            v[dst:dst + 2] = out
        elif op in ("v_add_f32_e64", "v_fma_f32", "v_mov_b32_e64"):
            a = [read(s[1], 1, at)[0] if s[0] == "v" else s[1] for s in srcs]
            v[dst] = a[0] + a[1] if op == "v_add_f32_e64" else (a[0] * a[1] + a[2] if op == "v_fma_f32" else a[0])
        else:
            raise ValueError("unknown instruction: " + ln)
    assert not pending, "LDS reads still in flight at the end of the block"
    # ---- the flush's combine (bas_fused_split.hip)
    A = v[0:32].reshape(16, 2)
    B = np.vstack([v[32:64].reshape(16, 2), v[(114 if psplit else 96):(116 if psplit else 98)].reshape(1, 2)])   # B[p-1], p = 0 .. 16
    if psplit:
        PA, PB, PP = v[64:80].reshape(8, 2), v[80:98].reshape(9, 2), v[98:114].reshape(8, 2)
        P = np.empty((16, 2))
        for r in range(8):
            P[2 * r] = PA[r] + PB[r]                           # PB[r] holds PB_{r-1}
            P[2 * r + 1] = (PP[r] - PA[r]) - PB[r + 1]
    else:
        P = v[64:96].reshape(16, 2)
    y = np.empty((32, 2))
    for p in range(16):
        y[2 * p] = A[p] + B[p]
        y[2 * p + 1] = (P[p] - A[p]) - B[p + 1]
    # ---- the definition
    want = np.zeros((32, 2))
    for r in range(n_steps):
        for a_ in range(32):
            u = a_ // (32 // nsub)
            w = al[r] if nsub == 1 else (al[r] if u == 0 else bl[r]) if nsub == 2 else al[r] + u * dl
            for o in range(32):
                t = o - a_ + 32
                k = 32 * r - 32 + t
                if 0 <= k < lseg and 0 <= t < 64:
                    want[o] += x[r, a_] * (h0[r, t] + w * d[r, t])
    return y, want
