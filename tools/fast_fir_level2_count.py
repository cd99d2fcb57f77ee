#!/usr/bin/env python3
"""Second level of the 2-parallel fast FIR, counted exactly (decide with numbers): packed vector
instructions and registers of one row step - a block of R outputs x 32 inputs of one lane, taps -31 .. R - 1 + ... - for
  level 1   (shipped): y_e = A + z^-1 B, y_o = P - A - B with A = g_e*x_e, B = g_o*x_o, P = (g_e+g_o)*(x_e+x_o)
  level 2   every half-rate product split once more the same way (9 quarter-rate products)
  level 1.5 only ONE of the three half-rate products split (what 32 more registers - the rolling x row - would buy)
for rows of R = 32 outputs (the shipped row) and R = 16 (what would let two filter waves share a SIMD).
Counted: FIR FMAs (one v_pk_fma_f32 per (output pair entry, tap) of every product, both ears packed), crossfade forming
(one v_pk_fma_f32 per full-rate tap the row step touches, apply_hrtf.py:442-443), tap sums (v_pk_add_f32), input sums
(v_add_f32), accumulator / operand registers.  Time model: profiles/r04_ab_sensitivity.txt (kernel time follows the
vector-instruction count with slope 0.65 at 3 605 instructions per unit, a tap read costs 1.83 instructions)."""


def conv_pairs(n_out, n_in, lo):
    """number of (output p, input q) pairs of a product with outputs p in [0, n_out), inputs q in [0, n_in), taps p - q >= lo"""
    return sum(1 for p in range(n_out) for q in range(n_in) if p - q >= lo)


def level(r_out, n_in, depth, split=(True, True, True)):
    """(fma, acc_entries, tap_seqs, x_seqs) of a Toeplitz block r_out x n_in evaluated with `depth` polyphase levels; every
    tap between an output and an EARLIER-or-equal input of the row pair is live (a full block: all r_out x n_in pairs)"""
    if depth == 0:
        return r_out * n_in, r_out, 1, 1
    ho, hi = (r_out + 1) // 2, n_in // 2
    f = a = t = x = 0
    for k, (o, i) in enumerate(((ho, hi), (ho + 1, hi), (ho, hi))):      # A, B (one more output entry: the z^-1), P
        d = depth - 1 if split[k] else 0
        ff, aa, tt, xx = level(o, i, d)
        f, a, t, x = f + ff, a + aa, t + tt, x + xx
    return f, a, t, x


def row_step(r_out, depth, split=(True, True, True)):
    n_in = 32
    taps = r_out + n_in - 1                                   # full-rate taps a row step touches
    fma, acc, tseq, xseq = level(r_out, n_in, depth, split)
    # forming: one FMA per full-rate tap (g = h0 + al d), then the sums of every level: a sequence at half the rate per sum
    t_adds = 0
    x_adds = 0
    if depth >= 1:
        t_adds += taps / 2                                    # g_e + g_o
        x_adds += n_in / 2
    if depth >= 2:
        n = sum(split)
        t_adds += n * taps / 4
        x_adds += n * n_in / 4
    regs_acc = 2 * acc
    regs_x = n_in * (1 if depth == 0 else (1.5 if depth == 1 else 1.5 + 0.75 * sum(split) / 3 * 1.0))
    return dict(fma=fma, form=taps, t_adds=t_adds, x_adds=x_adds, acc_regs=regs_acc, x_regs=regs_x, taps=taps,
                total=fma + taps + t_adds + x_adds)


def main():
    base = row_step(32, 1)
    print(__doc__)
    print(f"{'form':34s} {'FIR FMA':>8s} {'forming':>8s} {'tap sums':>9s} {'x sums':>7s} {'total':>7s} {'per 32x32':>10s} {'vs shipped':>11s} "
          f"{'acc regs':>9s} {'tap reads':>10s}")
    for name, r, d, sp in (("direct form, 32 outputs", 32, 0, (1, 1, 1)), ("level 1, 32 outputs (shipped)", 32, 1, (1, 1, 1)),
                           ("level 1.5 (P split), 32 outputs", 32, 2, (0, 0, 1)), ("level 2, 32 outputs", 32, 2, (1, 1, 1)),
                           ("level 1, 16 outputs", 16, 1, (1, 1, 1)), ("level 2, 16 outputs", 16, 2, (1, 1, 1))):
        s = row_step(r, d, tuple(bool(v) for v in sp))
        per = s["total"] * 32 / r
        reads = s["taps"] * 32 / r
        print(f"{name:34s} {s['fma']:8.0f} {s['form']:8.0f} {s['t_adds']:9.0f} {s['x_adds']:7.0f} {s['total']:7.0f} {per:10.0f} "
              f"{100 * (per / base['total'] - 1):+10.1f}% {s['acc_regs']:9.0f} {reads:10.0f}")
    print()
    print("(A full row step: the shipped generator emits 784 FIR FMAs + 64 forming FMAs + 32 tap sums + 16 x sums = 896 for 63 live")
    print(" taps; this count takes all 63 / 47 taps of a full block - the same to within the block's corner.)")
    print("Registers of the shipped unit block: 98 accumulators + 32 x + 16 sums + 48 taps + 12 formed + 2 = 208 (+ 11 operands).")
    print("Level 2 on 32-output rows: ~156 accumulators + 72 x-side + 18 formed + 32 taps = 278 > 256: does not exist.")
    print("Level 1.5: 116 accumulators + 56 x-side + 14 formed + 32 taps + 2 = 220 (+ 11): fits since the rolling x row, and")
    print("  saves 3.6 % of the vector instructions = 2.3 % of the kernel time by the measured slope.  BUILT (ffa_unitp_asm, the")
    print("  unit blocks of subchunks >= 32: 3 485 instead of 3 605 instructions per unit with the block's corners): -2.5 % kernel")
    print("  time measured (profiles/r04_ab_psplit.txt) - the slope's prediction.")
    print("Level 2 on 16-output rows (84 accumulators: two filter waves per SIMD would fit): forming and tap sums per output")
    print("  double, so a 32 x 32 block costs 7.7 % MORE vector instructions than the shipped form, and 1.5 x the tap reads (a tap")
    print("  read costs the kernel what 1.83 vector instructions do): slower by the measured prices.  Not built.")


if __name__ == "__main__":
    main()
