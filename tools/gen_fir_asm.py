#!/usr/bin/env python3
"""Generator of binaural-audio-synthesis_amd/csrc/bas_fir_asm.inc: the row step of the fused FIR kernel
(bas_fir.h: ffa_row_step_x, the 2-parallel fast FIR on a 32 x 32 Toeplitz block of packed FMAs) as ONE hand-scheduled
gfx950 assembly block with explicit VGPR numbers.

Why assembly: a wave that has its SIMD to itself - its CU partner is staging - waits
for the 8 tap reads of every octet before the octet's 98 FMAs (~25 % of its cycles), and hipcc cannot be made to issue
the reads of octet I + 1 under the FMAs of octet I without spilling hundreds of registers.  Here the taps are double
buffered in fixed registers, and every 64-bit instruction starts at an address = 0 mod 8 (an instruction at = 4 mod 8
costs one more issue cycle: profiles/r03_ubench_bank_placement.txt).

Register map (per lane; the accumulators are C++ variables pinned to these registers by "+{v[a:b]}" constraints):
    v[0:31]    A[p] = g_e * x_e       16 pairs (left, right)         v[98:129]   x row: 32 inputs
    v[32:63]   B[p-1] = g_o * x_o     entries 0..15                  v[130:145]  x_e + x_o: 16 sums
    v[64:95]   P[p] = (g_e+g_o)*(x_e+x_o)                            v[146:161]  taps: 4 x (h0_L, h0_R, d_L, d_R)
    v[96:97]   B entry 16                                            v[162:165] g_e[2]  v[166:169] g_o[2]  v[170:173] g_e + g_o
                                                                     v[174:175]  crossfade weight (low half)
Inputs: LDS byte address of the lane's x row, of its tap row (tap - 32 of the row distance), the weight, and the mask of
live octets (SGPR).  Arithmetic, order of accumulation per accumulator and results are those of ffa_octet_fma.

    python3 tools/gen_fir_asm.py [XR ...]    (XR = column stride of the x image in float4 units = rows + 1; default: 261 69,
                                              the tiles of 8192 and 2048 outputs)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "binaural-audio-synthesis_amd", "csrc", "bas_fir_asm.inc")

A0, B0, P0, B16 = 0, 32, 64, 96
X0, XS0 = 98, 130
TAP = 146                       # ONE buffer of 4 taps (16 registers): it is dead once the taps are formed
GE0, GO0, GS0 = 162, 166, 170
AL = 174
LAST = 175
NOWAIT = NOBRANCH = NOTAPS = NOALIGN = NOX = GENERIC_ONLY = False


def pr(r):
    return f"v[{r}:{r + 1}]"


def quad(r):
    return f"v[{r}:{r + 3}]"


def gen(xr_stride):
    """The schedule works in HALF octets (4 taps = 16 registers): forming the crossfaded taps consumes the tap buffer, so
    the reads of half u + 1 go out into the SAME buffer right behind the forming of half u and land under its ~49 FMAs.
    (A whole-octet double buffer needs 60 more registers per lane; those came back as scratch traffic in the staging code
    around the block: 16 % slower than hipcc's schedule.  Two half-octet buffers: 4 % slower.)"""
    L = []

    def emit(ln):
        if NOWAIT and ln.startswith("s_waitcnt"):
            return
        if NOBRANCH and (ln.startswith("s_bitcmp") or ln.startswith("s_cbranch") or ln.startswith("s_branch")):
            return
        if NOTAPS and ln.startswith("ds_read") and "%[tap]" in ln:
            return
        if NOX and ln.startswith("ds_read") and "%[xrow]" in ln:
            return
        if NOALIGN and ln.startswith(".p2align"):
            return
        L.append(ln)

    def align():
        emit(".p2align 3")

    def load_taps(u):
        i, h = u >> 1, u & 1
        for j in range(4):
            emit(f"ds_read_b128 {quad(TAP + 4 * j)}, %[tap] offset:{(8 * i + 4 * h + j) * 16}")

    def form(u):
        # crossfaded taps of the half octet: g = h0 + al d; even full-rate taps -> g_e, odd -> g_o, and their sums
        for k in range(2):
            te, to = TAP + 4 * (2 * k), TAP + 4 * (2 * k + 1)
            emit(f"v_pk_fma_f32 {pr(GE0 + 2 * k)}, {pr(te + 2)}, {pr(AL)}, {pr(te)} op_sel_hi:[1,0,1]")
            emit(f"v_pk_fma_f32 {pr(GO0 + 2 * k)}, {pr(to + 2)}, {pr(AL)}, {pr(to)} op_sel_hi:[1,0,1]")
        for k in range(2):
            emit(f"v_pk_add_f32 {pr(GS0 + 2 * k)}, {pr(GE0 + 2 * k)}, {pr(GO0 + 2 * k)}")

    def fmas(u):
        i, h = u >> 1, u & 1
        for k in range(2):
            jj = 2 * h + k
            dk = 4 * i + jj - 16
            for p in range(-1, 16):
                q = p - dk
                if not (0 <= q < 16):
                    continue
                xp = pr(X0 + 2 * q)                                  # (x_e[q], x_o[q]) = inputs 2q, 2q + 1
                xsp, xs_hi = pr(XS0 + 2 * (q >> 1)), q & 1
                if p >= 0:
                    emit(f"v_pk_fma_f32 {pr(A0 + 2 * p)}, {pr(GE0 + 2 * k)}, {xp}, {pr(A0 + 2 * p)} op_sel_hi:[1,0,1]")
                    sel = "op_sel:[0,1,0]" if xs_hi else "op_sel_hi:[1,0,1]"
                    emit(f"v_pk_fma_f32 {pr(P0 + 2 * p)}, {pr(GS0 + 2 * k)}, {xsp}, {pr(P0 + 2 * p)} {sel}")
                breg = B16 if p + 1 == 16 else B0 + 2 * (p + 1)
                emit(f"v_pk_fma_f32 {pr(breg)}, {pr(GO0 + 2 * k)}, {xp}, {pr(breg)} op_sel:[0,1,0]")

    def prologue(first_live):
        """weight into its pair, the x row, taps of the first half octet; first_live: None = by the mask (octet 0), else the
        number of the first live octet (straight-line variants)"""
        align()
        emit(f"v_mov_b32_e64 v{AL}, %[al]")
        for c in range(8):
            emit(f"ds_read_b128 {quad(X0 + 4 * c)}, %[xrow] offset:{c * xr_stride * 16}")
        if first_live is None:
            emit("s_bitcmp1_b32 %[mask], 0")
            emit("s_cbranch_scc0 L_x_only_%=")
            load_taps(0)
            emit("L_x_only_%=:")
        else:
            load_taps(2 * first_live)
        emit("s_waitcnt lgkmcnt(0)")
        align()
        for q in range(16):
            emit(f"v_add_f32_e64 v{XS0 + q}, v{X0 + 2 * q}, v{X0 + 2 * q + 1}")

    def straight(lo, hi, tag):
        """octets lo .. hi - 1 live, known at assembly time: no mask tests, no branches"""
        emit(f"L_{tag}_%=:")
        prologue(lo)
        for i in range(lo, hi):
            emit("s_waitcnt lgkmcnt(0)")
            align()
            form(2 * i)
            load_taps(2 * i + 1)
            fmas(2 * i)
            emit("s_waitcnt lgkmcnt(0)")
            align()
            form(2 * i + 1)
            if i + 1 < hi:
                load_taps(2 * i + 2)
            fmas(2 * i + 1)

    # ---- the masks of a 128-tap segment (0xff: rows 1-3, 0xf0: row 0, 0x0f: row 4) run straight-line code; any other
    # mask (shorter segments) takes the general block below.  A wave that has its SIMD to itself pays ~6 clocks for every
    # scalar instruction (profiles/r03_ubench_lone_wave.txt): the mask tests and branches were 9 % of a row step.
    if not NOBRANCH and not GENERIC_ONLY:
        for m, tag in ((0xff, "ff"), (0xf0, "f0"), (0x0f, "0f")):
            emit(f"s_cmp_eq_u32 %[mask], {m}")
            emit(f"s_cbranch_scc1 L_{tag}_%=")
    # ---- general block: octets by the mask (both halves of an octet are live or dead together)
    prologue(None)
    for i in range(8):
        emit(f"s_bitcmp1_b32 %[mask], {i}")
        emit(f"s_cbranch_scc0 L_dead{i}_%=")
        emit("s_waitcnt lgkmcnt(0)")                        # (its reads went out under the previous half's FMAs)
        align()
        form(2 * i)
        load_taps(2 * i + 1)
        fmas(2 * i)
        emit("s_waitcnt lgkmcnt(0)")
        align()
        form(2 * i + 1)
        if i < 7:
            emit(f"s_bitcmp1_b32 %[mask], {i + 1}")
            emit(f"s_cbranch_scc0 L_last{i}_%=")
            load_taps(2 * i + 2)
            emit(f"L_last{i}_%=:")
            align()
        fmas(2 * i + 1)
        if NOBRANCH:
            continue
        if i < 7:
            emit(f"s_branch L_end{i}_%=")
            emit(f"L_dead{i}_%=:")
            emit(f"s_bitcmp1_b32 %[mask], {i + 1}")
            emit(f"s_cbranch_scc0 L_end{i}_%=")
            load_taps(2 * i + 2)
            emit(f"L_end{i}_%=:")
        else:
            emit(f"L_dead{i}_%=:")
    if not NOBRANCH and not GENERIC_ONLY:
        emit("s_branch L_done_%=")
        straight(0, 8, "ff")
        emit("s_branch L_done_%=")
        straight(4, 8, "f0")
        emit("s_branch L_done_%=")
        straight(0, 4, "0f")
        emit("L_done_%=:")
    return L


# ---------------------------------------------------------------------------------------------------------------------------
# The whole (tile, source) unit of a 128-tap segment as ONE block (split-role kernel, bas_fused_split.hip: the filter waves
# have no staging state to keep, so the block may hold 126 operand registers).  Five row steps rp = 0..4 with the octets
# 4-7 | 0-7 | 0-7 | 0-7 | 0-3 live = 32 octets in a row, and what the per-step block cannot do:
#   * the x row of step r + 1 is read into the OTHER x buffer under the last octet of step r, its first taps likewise: no
#     exposed LDS latency at a step's start (5 x ~200 clocks per unit for a wave that has its SIMD to itself);
#   * taps are read one whole octet ahead into two half-octet buffers, so every wait is an lgkmcnt(N > 0) that is met;
#   * no mask tests, no branches, no per-step address arithmetic between the steps (tap address and weight of every step are
#     operands).
U_XA, U_XB, U_XS = 98, 130, 162
U_T = (178, 194)
U_GE, U_GO, U_GS = 210, 214, 218
U_AL = 222
U_LAST = 223
U_LSEGS = (128, 104)            # segment lengths with a unit block: L = 121 .. 128, and 97 .. 104 (the reference's default is 100)
ONE_WAIT = True


def unit_steps(lseg):
    """live octets [lo, hi) of the row steps rp = 0 .. halo of a segment of lseg taps (a multiple of 8): octet i of step rp
    holds taps 32 rp - 32 + 8 i .. + 7"""
    halo = (lseg + 31) >> 5
    return tuple((max(0, 4 - 4 * r), min(8, (lseg + 32 - 32 * r) >> 3)) for r in range(halo + 1))


NOFORM = False
ROLL = 3                        # unit block: rolling x row with this many half-octet tap buffers (0: round 3's two x buffers)
FMA_KEEP = 8
SPREAD = 0                      # unit block: LDS reads dealt one by one over the first SPREAD/8 of the FMA run behind them (0: in a burst)


def gen_unit(xr_stride, lseg):
    U_STEPS = unit_steps(lseg)
    halo = len(U_STEPS) - 1
    L = []

    def emit(ln):
        if NOWAIT and ln.startswith("s_waitcnt") and L:
            return
        if NOFORM and "%[al" not in ln and (f"{pr(U_AL)}," in ln or ln.startswith("v_pk_add_f32")):
            return                                              # (sensitivity run: no crossfade forming - wrong results)
        if FMA_KEEP < 8 and ln.startswith("v_pk_fma_f32") and f"{pr(U_AL)}," not in ln:
            emit.n = getattr(emit, "n", 0) + 1
            if emit.n % 8 >= FMA_KEEP:
                return                                          # (sensitivity run: FMA_KEEP of every 8 FIR FMAs - wrong results)
        if NOTAPS and ln.startswith("ds_read") and "%[tap" in ln:
            return
        if NOX and ln.startswith("ds_read") and "%[xrow]" in ln:
            return
        if NOALIGN and ln.startswith(".p2align"):
            return
        L.append(ln)
    queue = []                                                  # tags of the LDS reads in flight, oldest first

    def align():
        emit(".p2align 3")

    def issue(tag, line):
        emit(line)
        queue.append(tag)

    def wait_for(*tags):
        """LDS reads return in order: wait until no read with one of these tags is in flight"""
        if not any(t in tags for t in queue):
            return
        last = max(k for k, t in enumerate(queue) if t in tags)
        n = len(queue) - 1 - last
        assert n <= 15
        emit(f"s_waitcnt lgkmcnt({n})")
        del queue[:last + 1]
        align()

    def x_reads(r):
        xb = U_XA if r % 2 == 0 else U_XB
        return [(f"x{r}", f"ds_read_b128 {quad(xb + 4 * c)}, %[xrow] offset:{(halo - r) * 16 + c * xr_stride * 16}") for c in range(8)]

    def tap_reads(r, i, h):
        return [(f"t{r}.{i}.{h}", f"ds_read_b128 {quad(U_T[h] + 4 * j)}, %[tap{r}] offset:{(8 * i + 4 * h + j) * 16}") for j in range(4)]

    def form(h):
        tb = U_T[h]
        for k in range(2):
            te, to = tb + 4 * (2 * k), tb + 4 * (2 * k + 1)
            emit(f"v_pk_fma_f32 {pr(U_GE + 2 * k)}, {pr(te + 2)}, {pr(U_AL)}, {pr(te)} op_sel_hi:[1,0,1]")
            emit(f"v_pk_fma_f32 {pr(U_GO + 2 * k)}, {pr(to + 2)}, {pr(U_AL)}, {pr(to)} op_sel_hi:[1,0,1]")
        for k in range(2):
            emit(f"v_pk_add_f32 {pr(U_GS + 2 * k)}, {pr(U_GE + 2 * k)}, {pr(U_GO + 2 * k)}")

    def fmas(xb, i, h, reads):
        """the FMAs of half octet (i, h); `reads` (tag, line) go out in front of them (SPREAD = 0) or one by one between them"""
        lines = []
        for k in range(2):
            jj = 2 * h + k
            dk = 4 * i + jj - 16
            for p in range(-1, 16):
                q = p - dk
                if not (0 <= q < 16):
                    continue
                xp = pr(xb + 2 * q)
                xsp, xs_hi = pr(U_XS + 2 * (q >> 1)), q & 1
                if p >= 0:
                    lines.append(f"v_pk_fma_f32 {pr(A0 + 2 * p)}, {pr(U_GE + 2 * k)}, {xp}, {pr(A0 + 2 * p)} op_sel_hi:[1,0,1]")
                    sel = "op_sel:[0,1,0]" if xs_hi else "op_sel_hi:[1,0,1]"
                    lines.append(f"v_pk_fma_f32 {pr(P0 + 2 * p)}, {pr(U_GS + 2 * k)}, {xsp}, {pr(P0 + 2 * p)} {sel}")
                breg = B16 if p + 1 == 16 else B0 + 2 * (p + 1)
                lines.append(f"v_pk_fma_f32 {pr(breg)}, {pr(U_GO + 2 * k)}, {xp}, {pr(breg)} op_sel:[0,1,0]")
        reads = list(reads)
        if not SPREAD or not reads:
            for t, ln in reads:
                issue(t, ln)
            for ln in lines:
                emit(ln)
            return
        span = max(len(lines) * SPREAD // 8, len(reads))       # FMAs the reads are dealt over
        at = {(k * span) // len(reads): None for k in range(len(reads))}
        for n_, ln in enumerate(lines):
            if n_ in at and reads:
                issue(*reads.pop(0))
            emit(ln)
        for t, ln in reads:
            issue(t, ln)

    octets = [(r, i) for r, (lo, hi) in enumerate(U_STEPS) for i in range(lo, hi)]
    emit("s_waitcnt lgkmcnt(0)")                               # (scalar loads of the code around the block return out of order)
    align()
    for t, ln in x_reads(0) + tap_reads(*octets[0], 0) + tap_reads(*octets[0], 1):
        issue(t, ln)
    for n, (r, i) in enumerate(octets):
        xb = U_XA if r % 2 == 0 else U_XB
        nxt = octets[n + 1] if n + 1 < len(octets) else None
        if i == U_STEPS[r][0]:                                  # first octet of a step: its x row, the sums, the weight
            wait_for(f"x{r}")
            for q in range(16):
                emit(f"v_add_f32_e64 v{U_XS + q}, v{xb + 2 * q}, v{xb + 2 * q + 1}")
            emit(f"v_mov_b32_e64 v{U_AL}, %[al{r}]")
        last_of_step = i == U_STEPS[r][1] - 1
        for h in range(2):
            if ONE_WAIT:
                # ONE wait per octet, in its middle: for the octet's second half (read a whole octet ago) and the NEXT octet's
                # first half (read half an octet ago, behind the forming of this octet's first half)
                if h == 1:
                    wait_for(f"t{r}.{i}.1", *([f"t{nxt[0]}.{nxt[1]}.0"] if nxt else []))
                elif n == 0:
                    wait_for(f"t{r}.{i}.0")
            else:
                wait_for(f"t{r}.{i}.{h}")
            form(h)
            reads = []
            if h == 0 and last_of_step and r + 1 < len(U_STEPS):
                reads += x_reads(r + 1)                         # (in front of the next octet's taps: it is needed first)
            if nxt:
                reads += tap_reads(*nxt, h)
            fmas(xb, i, h, reads)
    assert not queue
    return L


# ---------------------------------------------------------------------------------------------------------------------------
# The unit block with a ROLLING x row (round 4): the 8 quads of an x row are not used evenly - octet i of a row step
# meets inputs q in [12 - 4 i, 31 - 4 i] - so quad c of the NEXT row is read into the same registers as soon as the last
# octet that needs quad c of this row has gone out (quads 7, 6 behind octet 4, .. quads 1, 0 behind octet 7), three octets
# before its first use.  One x buffer instead of two: 32 registers free for
#   * NSUB = 2: subchunks of 16 samples - a row of 32 inputs then meets TWO crossfaded tap sets (inputs 0-15 with the
#     weight al, 16-31 with al + S / K; apply_hrtf.py:442-443), 12 more formed-tap registers; a half octet forms only the
#     sets its inputs fall into;
#   * NBUF = 3 or 4 half-octet tap buffers (reads one and a half or two octets ahead).
# Same arithmetic and order of accumulation per accumulator as gen_unit (bit-identical output for NSUB = 1).
R_X, R_XS, R_T0 = 98, 130, 146


def gen_unit_roll(xr_stride, lseg, nsub=1, nbuf=2, psplit=False):
    """psplit: the half-rate product P = (g_e + g_o) * (x_e + x_o) is split ONCE MORE the same way (a second fast-FIR level
    for one of the three products: what the registers of the rolling x row allow; profiles/r04_ubench_fast_fir_level2.txt):
    with xs = x_e + x_o, gs = g_e + g_o at half rate,  P[2r] = PA[r] + PB[r-1],  P[2r+1] = PP[r] - PA[r] - PB[r],
    PA = gs_even * xs_even, PB = gs_odd * xs_odd, PP = (gs_even + gs_odd) * (xs_even + xs_odd) at quarter rate: a half octet
    carries ONE quarter-rate tap (its two half-rate taps are the even and the odd one), so its 2 x 16 P FMAs become
    8 + 9 + 8 at the most.  Accumulators: A v[0:31], B v[32:63], PA v[64:79], PB v[80:97] (r = -1 .. 7), PP v[98:113],
    B entry 16 v[114:115]; the operands start at v116."""
    assert not (psplit and nsub != 1)
    U_STEPS = unit_steps(lseg)
    halo = len(U_STEPS) - 1
    R_X = 116 if psplit else 98                                 # (shadow the module's register map: 18 more accumulators)
    R_XS = R_X + 32
    R_XSS = R_XS + 16                                           # psplit: xs_even + xs_odd, 8 values
    R_T0 = R_XS + 16 + (8 if psplit else 0)
    PA0, PB0, PP0 = 64, 80, 98
    b16 = 114 if psplit else B16
    r_g = R_T0 + 16 * nbuf                                     # formed taps: nsub x (ge[2], go[2], gs[2]) pairs
    r_gss = r_g + 12 * nsub                                    # psplit: gs_even + gs_odd of the half octet
    r_al = r_gss + (2 if psplit else 0)                        # the crossfade weights of the row's nsub subchunks, two per pair
    last = r_al + (3 if nsub == 4 else 1)
    L = []

    def emit(ln):                                               # (sensitivity variants, wrong results: tools/build_fir_variants.sh)
        if NOWAIT and ln.startswith("s_waitcnt") and L:
            return
        if NOTAPS and ln.startswith("ds_read") and "%[tap" in ln:
            return
        if NOX and ln.startswith("ds_read") and "%[xrow]" in ln:
            return
        if NOALIGN and ln.startswith(".p2align"):
            return
        if NOFORM and "%[" not in ln and ("op_sel_hi:[1,1,1]" in ln or f", {pr(r_al)}, " in ln or f", {pr(r_al + 2)}, " in ln
                                          or ln.startswith("v_pk_add_f32")):
            return
        if FMA_KEEP < 8 and ln.startswith("v_pk_fma_f32") and f", {pr(r_al)}, " not in ln and f", {pr(r_al + 2)}, " not in ln:
            emit.n = getattr(emit, "n", 0) + 1
            if emit.n % 8 >= FMA_KEEP:
                return
        L.append(ln)
    queue = []

    def align():
        emit(".p2align 3")

    def issue(tag, line):
        emit(line)
        queue.append(tag)

    landed = set()                                              # tags of the reads a wait has covered

    def wait_for(tags):
        if not any(t in tags for t in queue):
            return
        last_ = max(k for k, t in enumerate(queue) if t in tags)
        n = len(queue) - 1 - last_
        assert n <= 15, n
        emit(f"s_waitcnt lgkmcnt({n})")
        landed.update(queue[:last_ + 1])
        del queue[:last_ + 1]
        align()

    halves = [(r, i, h) for r, (lo, hi) in enumerate(U_STEPS) for i in range(lo, hi) for h in range(2)]

    def fma_list(r, i, h):
        """(lines, quads read, tap sets used) of half octet (i, h)"""
        lines, quads, sets = [], set(), set()
        for k in range(2):
            jj = 2 * h + k
            dk = 4 * i + jj - 16
            for p in range(-1, 16):
                q = p - dk
                if not (0 <= q < 16):
                    continue
                u = q // (16 // nsub)                          # subchunk of the row that inputs 2 q, 2 q + 1 fall into
                sets.add(u)
                quads.add(q >> 1)
                ge, go, gs = r_g + 12 * u, r_g + 12 * u + 4, r_g + 12 * u + 8
                xp = pr(R_X + 2 * q)
                xsp, xs_hi = pr(R_XS + 2 * (q >> 1)), q & 1
                if p >= 0:
                    lines.append(f"v_pk_fma_f32 {pr(A0 + 2 * p)}, {pr(ge + 2 * k)}, {xp}, {pr(A0 + 2 * p)} op_sel_hi:[1,0,1]")
                    sel = "op_sel:[0,1,0]" if xs_hi else "op_sel_hi:[1,0,1]"
                    if not psplit:
                        lines.append(f"v_pk_fma_f32 {pr(P0 + 2 * p)}, {pr(gs + 2 * k)}, {xsp}, {pr(P0 + 2 * p)} {sel}")
                breg = b16 if p + 1 == 16 else B0 + 2 * (p + 1)
                lines.append(f"v_pk_fma_f32 {pr(breg)}, {pr(go + 2 * k)}, {xp}, {pr(breg)} op_sel:[0,1,0]")
        if psplit:                                              # the quarter-rate tap j of this half octet against xs_even / xs_odd / their sum
            j = 2 * i + h - 8
            gs = r_g + 8
            for r_ in range(-1, 8):
                t = r_ - j
                if not (0 <= t < 8):
                    continue
                quads.add(t)
                xs_pair = pr(R_XS + 2 * t)                      # (xs[2t], xs[2t+1]) = (xs_even[t], xs_odd[t])
                xss_pair, xss_hi = pr(R_XSS + 2 * (t >> 1)), t & 1
                if r_ >= 0:
                    lines.append(f"v_pk_fma_f32 {pr(PA0 + 2 * r_)}, {pr(gs)}, {xs_pair}, {pr(PA0 + 2 * r_)} op_sel_hi:[1,0,1]")
                    sel = "op_sel:[0,1,0]" if xss_hi else "op_sel_hi:[1,0,1]"
                    lines.append(f"v_pk_fma_f32 {pr(PP0 + 2 * r_)}, {pr(r_gss)}, {xss_pair}, {pr(PP0 + 2 * r_)} {sel}")
                lines.append(f"v_pk_fma_f32 {pr(PB0 + 2 * (r_ + 1))}, {pr(gs + 2)}, {xs_pair}, {pr(PB0 + 2 * (r_ + 1))} op_sel:[0,1,0]")
        return lines, quads, sets

    plan = [fma_list(*hv) for hv in halves]
    first_use, last_use = {}, {}
    for n, ((r, i, h), (_, quads, _)) in enumerate(zip(halves, plan)):
        for c in quads:
            first_use.setdefault((r, c), n)
            last_use[(r, c)] = n

    def x_read(r, c):
        return (f"x{r}.{c}", f"ds_read_b128 {quad(R_X + 4 * c)}, %[xrow] offset:{(halo - r) * 16 + c * xr_stride * 16}")

    def tap_reads(n):
        r, i, h = halves[n]
        b = R_T0 + 16 * (n % nbuf)
        return [(f"t{n}", f"ds_read_b128 {quad(b + 4 * j)}, %[tap{r}] offset:{(8 * i + 4 * h + j) * 16}") for j in range(4)]

    def form(n, sets):
        tb = R_T0 + 16 * (n % nbuf)
        for u in sorted(sets):
            ge, go, gs = r_g + 12 * u, r_g + 12 * u + 4, r_g + 12 * u + 8
            wsel = "op_sel_hi:[1,0,1]" if u % 2 == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]"   # low / high half of the weight pair
            wp = pr(r_al + 2 * (u >> 1))
            for k in range(2):
                te, to = tb + 4 * (2 * k), tb + 4 * (2 * k + 1)
                emit(f"v_pk_fma_f32 {pr(ge + 2 * k)}, {pr(te + 2)}, {wp}, {pr(te)} {wsel}")
                emit(f"v_pk_fma_f32 {pr(go + 2 * k)}, {pr(to + 2)}, {wp}, {pr(to)} {wsel}")
            for k in range(2):
                emit(f"v_pk_add_f32 {pr(gs + 2 * k)}, {pr(ge + 2 * k)}, {pr(go + 2 * k)}")
        if psplit:
            emit(f"v_pk_add_f32 {pr(r_gss)}, {pr(r_g + 8)}, {pr(r_g + 10)}")

    # x quads of step r + 1 go out right behind the half octet that uses quad c of step r for the last time
    x_after = {}                                                # half index n -> reads issued in front of the FMAs of half n
    for (r, c), n_first in sorted(first_use.items()):
        if r == 0:
            continue
        prev = [last_use[(rr, c)] for rr in range(r) if (rr, c) in last_use]
        x_after.setdefault((max(prev) + 1) if prev else 0, []).append(x_read(r, c))
    wait_points = [n for n, (r, i, h) in enumerate(halves) if n == 0 or h == 1] if ONE_WAIT else list(range(len(halves)))

    emit("s_waitcnt lgkmcnt(0)")                               # (scalar loads of the code around the block return out of order)
    align()
    for c in sorted(c for (r, c) in first_use if r == 0):
        issue(*x_read(0, c))
    for n in range(min(nbuf, len(halves))):
        for t in tap_reads(n):
            issue(*t)
    xs_done = set()

    def needs(m):
        rm = halves[m][0]
        return [f"t{m}"] + [f"x{rm}.{c}" for c in sorted(plan[m][1]) if first_use[(rm, c)] == m]

    def sums_of(m):
        """x_e + x_o of the quads half m is the first to use, once they have landed"""
        rm = halves[m][0]
        for c in sorted(plan[m][1]):
            if first_use[(rm, c)] == m and f"x{rm}.{c}" in landed and (rm, c) not in xs_done:
                xs_done.add((rm, c))
                for q in (2 * c, 2 * c + 1):
                    emit(f"v_add_f32_e64 v{R_XS + q}, v{R_X + 2 * q}, v{R_X + 2 * q + 1}")
                if psplit:
                    emit(f"v_add_f32_e64 v{R_XSS + c}, v{R_XS + 2 * c}, v{R_XS + 2 * c + 1}")

    for n, ((r, i, h), (lines, quads, sets)) in enumerate(zip(halves, plan)):
        nxt = min([m for m in wait_points if m > n], default=len(halves))
        if n in wait_points:                                    # what the halves up to the next wait point need, as far as it is on its way
            wait_for([t for m in range(n, nxt) for t in needs(m) if t in queue])
        missing = [t for t in needs(n) if t not in landed]
        if missing:                                             # (read too late for the wait point in front: a wait of its own)
            assert all(t in queue for t in missing), (n, missing)
            wait_for(missing)
        for m in range(n, nxt if n in wait_points else n + 1):
            sums_of(m)
        assert all(f"x{r}.{c}" in landed and (r, c) in xs_done for c in quads), (n, quads)
        if (i, h) == (U_STEPS[r][0], 0):                        # first half octet of a step: its weights
            emit(f"v_mov_b32_e64 v{r_al}, %[al{r}]")
            if nsub == 2:
                emit(f"v_mov_b32_e64 v{r_al + 1}, %[bl{r}]")
            if nsub == 4:                                       # al + u S / K for the row's subchunks u = 1, 2, 3
                emit(f"v_add_f32_e64 v{r_al + 1}, v{r_al}, %[dl]")
                emit(f"v_fma_f32 v{r_al + 2}, 2.0, %[dl], v{r_al}")
                emit(f"v_add_f32_e64 v{r_al + 3}, v{r_al + 2}, %[dl]")
        form(n, sets)
        reads = list(x_after.get(n + 1, []))                    # (quads whose last use is THIS half: behind its FMAs = in front of the next)
        pre = list(x_after.get(n, [])) if n == 0 else []
        nx = n + nbuf
        treads = tap_reads(nx) if nx < len(halves) else []
        for t in pre + treads:
            issue(*t)
        for ln in lines:
            emit(ln)
        for t in reads:
            issue(*t)
    assert not queue, queue
    return L, last


UNIT_DECL = """
template <int XR, int LSEG>
__device__ __forceinline__ void ffa_unit_asm(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                              const unsigned (&tap)[5], const float (&al)[5]);
"""

UNIT_FUNC = """
// One (tile, source) unit of a {lseg}-tap segment in one block (see tools/gen_fir_asm.py: gen_unit): {n_fma} v_pk_fma_f32,
// {n_ds} ds_read_b128, {n_wait} waits.  xrow4 = LDS address of the lane's x row of step 4 (the lowest: step r reads 16 (4 - r)
// bytes above); tap[r], al[r] = tap row address and crossfade weight of step r.
template <>
__device__ __forceinline__ void ffa_unit_asm<{xr}, {lseg}>(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                              const unsigned (&tap)[5], const float (&al)[5]) {{
    asm volatile(
{body}
        : "+{{v[0:31]}}"(accA), "+{{v[32:63]}}"(accB), "+{{v[64:95]}}"(accP), "+{{v[96:97]}}"(accB16)
        : [xrow] "v"(xrow4), [tap0] "v"(tap[0]), [tap1] "v"(tap[1]), [tap2] "v"(tap[2]), [tap3] "v"(tap[3]), [tap4] "v"(tap[4]),
          [al0] "v"(al[0]), [al1] "v"(al[1]), [al2] "v"(al[2]), [al3] "v"(al[3]), [al4] "v"(al[4])
        : "memory", {clob});
}}
"""


PSPLIT_NBUF = 2
UNITP_FUNC = """
// The same unit with the half-rate product P split once more (gen_unit_roll: psplit): {n_fma} v_pk_fma_f32, {n_ds} ds_read_b128,
// {n_wait} waits.  116 accumulator registers: A, B as before, PA v[64:79], PB v[80:97] (entries r = -1 .. 7), PP v[98:113],
// B entry 16 v[114:115]; the flush combines P[2r] = PA[r] + PB[r-1], P[2r+1] = PP[r] - PA[r] - PB[r] first.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int XR, int LSEG>
__device__ __forceinline__ void ffa_unitp_asm(f32x32 &accA, f32x32 &accB, f32x16 &accPA, f32x16 &accPB, f32x2 &accPB8, f32x16 &accPP,
                                               f32x2 &accB16, unsigned xrow4, const unsigned (&tap)[5], const float (&al)[5]);
template <>
__device__ __forceinline__ void ffa_unitp_asm<{xr}, {lseg}>(f32x32 &accA, f32x32 &accB, f32x16 &accPA, f32x16 &accPB, f32x2 &accPB8,
                                               f32x16 &accPP, f32x2 &accB16, unsigned xrow4, const unsigned (&tap)[5],
                                               const float (&al)[5]) {{
    asm volatile(
{body}
        : "+{{v[0:31]}}"(accA), "+{{v[32:63]}}"(accB), "+{{v[64:79]}}"(accPA), "+{{v[80:95]}}"(accPB), "+{{v[96:97]}}"(accPB8),
          "+{{v[98:113]}}"(accPP), "+{{v[114:115]}}"(accB16)
        : [xrow] "v"(xrow4), [tap0] "v"(tap[0]), [tap1] "v"(tap[1]), [tap2] "v"(tap[2]), [tap3] "v"(tap[3]), [tap4] "v"(tap[4]),
          [al0] "v"(al[0]), [al1] "v"(al[1]), [al2] "v"(al[2]), [al3] "v"(al[3]), [al4] "v"(al[4])
        : "memory", {clob});
}}
"""

UNIT2_FUNC = """
// The same unit for subchunks of 16 samples (apply_hrtf.py:401-402 accepts any divisor of the chunk; :442-443): inputs 0-15 of
// a row are crossfaded with al[r], inputs 16-31 with bl[r] - two formed tap sets, each formed only by the half octets whose
// inputs fall into it: {n_fma} v_pk_fma_f32, {n_ds} ds_read_b128, {n_wait} waits.
template <int XR, int LSEG>
__device__ __forceinline__ void ffa_unit2_asm(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                               const unsigned (&tap)[5], const float (&al)[5], const float (&bl)[5]);
template <>
__device__ __forceinline__ void ffa_unit2_asm<{xr}, {lseg}>(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                               const unsigned (&tap)[5], const float (&al)[5], const float (&bl)[5]) {{
    asm volatile(
{body}
        : "+{{v[0:31]}}"(accA), "+{{v[32:63]}}"(accB), "+{{v[64:95]}}"(accP), "+{{v[96:97]}}"(accB16)
        : [xrow] "v"(xrow4), [tap0] "v"(tap[0]), [tap1] "v"(tap[1]), [tap2] "v"(tap[2]), [tap3] "v"(tap[3]), [tap4] "v"(tap[4]),
          [al0] "v"(al[0]), [al1] "v"(al[1]), [al2] "v"(al[2]), [al3] "v"(al[3]), [al4] "v"(al[4]),
          [bl0] "v"(bl[0]), [bl1] "v"(bl[1]), [bl2] "v"(bl[2]), [bl3] "v"(bl[3]), [bl4] "v"(bl[4])
        : "memory", {clob});
}}
"""


UNIT4_FUNC = """
// The same unit for subchunks of 8 samples: four formed tap sets per row, each formed only by the half octets whose inputs
// fall into it; the weights of a row's subchunks are al[r] + u dl, dl = S / K (two tap buffers: the registers go to the
// formed sets): {n_fma} v_pk_fma_f32, {n_ds} ds_read_b128, {n_wait} waits.
template <int XR, int LSEG>
__device__ __forceinline__ void ffa_unit4_asm(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                               const unsigned (&tap)[5], const float (&al)[5], float dl);
template <>
__device__ __forceinline__ void ffa_unit4_asm<{xr}, {lseg}>(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow4,
                                               const unsigned (&tap)[5], const float (&al)[5], float dl) {{
    asm volatile(
{body}
        : "+{{v[0:31]}}"(accA), "+{{v[32:63]}}"(accB), "+{{v[64:95]}}"(accP), "+{{v[96:97]}}"(accB16)
        : [xrow] "v"(xrow4), [tap0] "v"(tap[0]), [tap1] "v"(tap[1]), [tap2] "v"(tap[2]), [tap3] "v"(tap[3]), [tap4] "v"(tap[4]),
          [al0] "v"(al[0]), [al1] "v"(al[1]), [al2] "v"(al[2]), [al3] "v"(al[3]), [al4] "v"(al[4]), [dl] "v"(dl)
        : "memory", {clob});
}}
"""


def main():
    check = "--check" in sys.argv[1:]                           # compare with the committed file instead of writing it
    # diagnostic variants (wrong results; tools/ubench_lone_wave.hip): --nowait no waits for the LDS reads, --nobranch no octet
    # masks (all live), --notaps no tap reads, --noalign no alignment padding, --nox no x-row reads; --out=FILE
    global NOWAIT, NOBRANCH, NOTAPS, NOALIGN, NOX, OUT, GENERIC_ONLY, ONE_WAIT, SPREAD, NOFORM, FMA_KEEP
    global ROLL
    NOFORM = "--noform" in sys.argv[1:]
    for a in sys.argv[1:]:
        if a.startswith("--roll="):                             # (A/B: the rolling-x unit block with this many tap buffers)
            ROLL = int(a[7:])
    for a in sys.argv[1:]:
        if a.startswith("--fma-keep="):
            FMA_KEEP = int(a[11:])
    ONE_WAIT = "--two-waits" not in sys.argv[1:]
    for a in sys.argv[1:]:
        if a.startswith("--spread="):                           # (A/B: LDS reads of the unit block dealt over the FMAs, in eighths of a run)
            SPREAD = int(a[9:])                # (A/B: the unit block with a wait in front of every half octet)
    GENERIC_ONLY = "--generic" in sys.argv[1:]                  # (A/B: the round-3 block before the straight-line variants)
    NOWAIT, NOBRANCH, NOTAPS = ("--" + k in sys.argv[1:] for k in ("nowait", "nobranch", "notaps"))
    NOALIGN, NOX = ("--" + k in sys.argv[1:] for k in ("noalign", "nox"))
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            OUT = a[6:]
    xrs = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [261, 69]   # tiles of 8192 (4 waves) and 2048 (1 wave): 260 / 68 rows + 1
    clob = ", ".join(f'"v{r}"' for r in range(X0, LAST + 1))
    text = None
    for xr in xrs:
        lines = gen(xr)
        body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
        n_fma = sum(1 for ln in lines if ln.startswith("v_pk_fma"))
        if text is None:
            text = HEAD.format(n_fma=n_fma, xrs=" ".join(str(x) for x in xrs))
        text += FUNC.format(xr=xr, body=body, clob=clob)
    if 261 in xrs and not GENERIC_ONLY:
        text += UNIT_DECL
    for lseg in (U_LSEGS if 261 in xrs and not GENERIC_ONLY else ()):
        if ROLL:
            ul, u_last = gen_unit_roll(261, lseg, 1, ROLL)
        else:
            ul, u_last = gen_unit(261, lseg), U_LAST
        text += UNIT_FUNC.format(xr=261, lseg=lseg, body="\n".join(f'        "{ln}\\n\\t"' for ln in ul),
                                 clob=", ".join(f'"v{r}"' for r in range(U_XA, u_last + 1)),
                                 n_fma=sum(1 for ln in ul if ln.startswith("v_pk_fma")),
                                 n_ds=sum(1 for ln in ul if ln.startswith("ds_read")),
                                 n_wait=sum(1 for ln in ul if ln.startswith("s_waitcnt")))
    for lseg in (U_LSEGS if 261 in xrs and not GENERIC_ONLY else ()):     # the P product split once more (fast-FIR level 1.5)
        ul, u_last = gen_unit_roll(261, lseg, 1, PSPLIT_NBUF, psplit=True)
        text += UNITP_FUNC.format(xr=261, lseg=lseg, body="\n".join(f'        "{ln}\\n\\t"' for ln in ul),
                                  clob=", ".join(f'"v{r}"' for r in range(116, u_last + 1)),
                                  n_fma=sum(1 for ln in ul if ln.startswith("v_pk_fma")),
                                  n_ds=sum(1 for ln in ul if ln.startswith("ds_read")),
                                  n_wait=sum(1 for ln in ul if ln.startswith("s_waitcnt")))
    for lseg in (U_LSEGS if 261 in xrs and not GENERIC_ONLY else ()):     # subchunks of 16: two tap sets per row
        ul, u_last = gen_unit_roll(261, lseg, 2, ROLL or 3)
        text += UNIT2_FUNC.format(xr=261, lseg=lseg, body="\n".join(f'        "{ln}\\n\\t"' for ln in ul),
                                  clob=", ".join(f'"v{r}"' for r in range(U_XA, u_last + 1)),
                                  n_fma=sum(1 for ln in ul if ln.startswith("v_pk_fma")),
                                  n_ds=sum(1 for ln in ul if ln.startswith("ds_read")),
                                  n_wait=sum(1 for ln in ul if ln.startswith("s_waitcnt")))
    for lseg in (U_LSEGS if 261 in xrs and not GENERIC_ONLY else ()):     # subchunks of 8: four tap sets per row
        ul, u_last = gen_unit_roll(261, lseg, 4, 2)
        text += UNIT4_FUNC.format(xr=261, lseg=lseg, body="\n".join(f'        "{ln}\\n\\t"' for ln in ul),
                                  clob=", ".join(f'"v{r}"' for r in range(U_XA, u_last + 1)),
                                  n_fma=sum(1 for ln in ul if ln.startswith("v_pk_fma")),
                                  n_ds=sum(1 for ln in ul if ln.startswith("ds_read")),
                                  n_wait=sum(1 for ln in ul if ln.startswith("s_waitcnt")))
    if check:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print(f"{OUT}: {'up to date' if same else 'DIFFERS from what the generator writes'}")
        sys.exit(0 if same else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {len(lines)} lines per variant, {n_fma} packed FMAs")


HEAD = """// GENERATED by tools/gen_fir_asm.py {xrs} - do not edit (see that file for the register map and the schedule).
// Row step of the fused FIR kernel as one gfx950 assembly block: {n_fma} v_pk_fma_f32 (784 FIR + 64 forming), 32 v_pk_add_f32,
// 16 v_add_f32, 72 ds_read_b128; reads of half octet u + 1 issued right behind the forming of half u, under its FMAs; every
// 64-bit instruction 8-byte aligned.  Same arithmetic and accumulation order as ffa_row_step_x (bas_fir.h).
// One instantiation per column stride XR of the x image (float4 units).
#pragma once
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int XR>
__device__ __forceinline__ void ffa_row_step_asm(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow,
                                                  unsigned tap, float al, unsigned mask);
"""

FUNC = """
template <>
__device__ __forceinline__ void ffa_row_step_asm<{xr}>(f32x32 &accA, f32x32 &accB, f32x2 &accB16, f32x32 &accP, unsigned xrow,
                                                  unsigned tap, float al, unsigned mask) {{
    asm volatile(
{body}
        : "+{{v[0:31]}}"(accA), "+{{v[32:63]}}"(accB), "+{{v[64:95]}}"(accP), "+{{v[96:97]}}"(accB16)
        : [xrow] "v"(xrow), [tap] "v"(tap), [al] "v"(al), [mask] "s"(mask)
        : "memory", "scc", {clob});
}}
"""


if __name__ == "__main__":
    main()
