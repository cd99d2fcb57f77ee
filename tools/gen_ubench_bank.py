#!/usr/bin/env python3
"""Generator of tools/ubench_bank_gen.hip: the FIR row step's v_pk_fma_f32 stream written as gfx950 assembly with
EXPLICIT VGPR numbers, to find where the 4.45 cycles per packed FMA of the compiler-allocated stream go
(DESIGN.md section 4.1).

Every variant is one kernel whose body is a single asm block:
    lane-varying operands loaded from `in` -> s_memtime -> LOOP { N vector instructions } -> s_memtime -> checksum
The host side (same file) runs every variant on zero and on random operands at 1 and 2 waves per SIMD and prints
cycles per instruction per wave (s_memtime ticks = shader cycles, median over workgroups), the in-kernel clock
(s_memtime / s_memrealtime) and the TFLOP/s of the launch.

    python3 tools/gen_ubench_bank.py && hipcc -O3 --offload-arch=gfx950 tools/ubench_bank_gen.hip -o tools/ubench_bank
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# ---------------------------------------------------------------------------------------------------------------
# register maps: a variant is (name, description, list of instruction strings, set of VGPRs used, flop per instr)
# ---------------------------------------------------------------------------------------------------------------


def pair(r):
    return f"v[{r}:{r + 1}]"


def fir_pattern(acc_regs, g_regs, x_regs, form="opsel", n_g=8):
    """The row step's pattern: for each formed tap g[j], 32 accumulators meet 32 inputs (a = (o + j) mod 32).
    acc_regs: first register of each accumulator pair; g_regs: of each tap pair; x_regs: first register of each
    pair of inputs (opsel form: x[a] = half a & 1 of pair a >> 1; dup form: pair a holds (x, x))."""
    ins = []
    na = len(acc_regs)
    for j in range(n_g):
        g = g_regs[j % len(g_regs)]
        for o in range(na):
            a = (o + 3 * j) % 32
            acc = acc_regs[o]
            if form == "opsel":
                xp = x_regs[(a >> 1) % len(x_regs)]
                sel = "op_sel_hi:[1,0,1]" if (a & 1) == 0 else "op_sel:[0,1,0]"
                ins.append(f"v_pk_fma_f32 {pair(acc)}, {pair(g)}, {pair(xp)}, {pair(acc)} {sel}")
            elif form == "dup":
                xp = x_regs[a % len(x_regs)]
                ins.append(f"v_pk_fma_f32 {pair(acc)}, {pair(g)}, {pair(xp)}, {pair(acc)}")
            elif form == "fma2":                       # two single-rate FMAs per packed one
                xp = x_regs[(a >> 1) % len(x_regs)] + (a & 1)
                ins.append(f"v_fma_f32 v{acc}, v{g}, v{xp}, v{acc}")
                ins.append(f"v_fma_f32 v{acc + 1}, v{g + 1}, v{xp}, v{acc + 1}")
            elif form == "pkmul":                      # no accumulator read: 2 source operands only
                xp = x_regs[(a >> 1) % len(x_regs)]
                ins.append(f"v_pk_mul_f32 {pair(acc)}, {pair(g)}, {pair(xp)} op_sel_hi:[1,0]")
            elif form == "pkadd":
                ins.append(f"v_pk_add_f32 {pair(acc)}, {pair(g)}, {pair(acc)}")
            else:
                raise ValueError(form)
    return ins


def regs_of(ins):
    import re
    used = set()
    for s in ins:
        for m in re.finditer(r"v\[(\d+):(\d+)\]", s):
            used.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", s):
            used.add(int(m.group(1)))
    return used


VARIANTS = []


def add(name, desc, ins, flop_per_ins=4 * 64):
    VARIANTS.append((name, desc, ins, flop_per_ins))


def build_variants():
    # natural layout: what a compiler would do - accumulators, taps and inputs in consecutive pairs
    acc32 = [64 + 2 * o for o in range(32)]           # v64..v127
    g8 = [48 + 2 * j for j in range(8)]               # v48..v63
    x16 = [16 + 2 * q for q in range(16)]             # v16..v47 (32 inputs as 16 pairs)
    xdup = [128 + 2 * a for a in range(32)]           # v128..v191: 32 duplicated inputs
    add("nat_opsel", "32 acc pairs consecutive, 8 taps, x broadcast by op_sel (the kernel's form)",
        fir_pattern(acc32, g8, x16, "opsel"))
    add("nat_dup", "same, x held as (x, x) pairs - no op_sel (what hipcc did in r02's ubench)",
        fir_pattern(acc32, g8, xdup, "dup"))
    add("nat_fma2", "same work as two v_fma_f32 per packed FMA", fir_pattern(acc32, g8, x16, "fma2"), 2 * 64)
    add("nat_pkmul", "v_pk_mul_f32 (two source operands)", fir_pattern(acc32, g8, x16, "pkmul"), 2 * 64)
    add("nat_pkadd", "v_pk_add_f32 acc += g (two source operands, no op_sel)", fir_pattern(acc32, g8, x16, "pkadd"),
        2 * 64)

    # bank placement: every operand pair of every instruction in a chosen residue class mod 4 (banks = reg mod 4;
    # a 64-bit pair covers banks {0,1} or {2,3})
    for ab in (0, 2):
        for gb in (0, 2):
            for xb in (0, 2):
                acc = [64 + 4 * o + ab for o in range(32)]          # v64..v191 step 4
                g = [192 + 4 * j + gb for j in range(8)]             # v192..v223
                x = [0 + 4 * q + xb for q in range(16)]              # v0..v63
                add(f"bank_a{ab}g{gb}x{xb}", f"acc pairs = {ab} mod 4, tap pairs = {gb} mod 4, x pairs = {xb} mod 4",
                    fir_pattern(acc, g, x, "opsel"))
    # the same sweep for the op_sel-free form (x as duplicated pairs)
    for gb in (0, 2):
        for xb in (0, 2):
            acc = [64 + 4 * o for o in range(32)]
            g = [192 + 4 * j + gb for j in range(8)]
            x = [0 + 4 * q + xb for q in range(16)]
            add(f"bankdup_a0g{gb}x{xb}", f"dup form: acc = 0, taps = {gb}, x = {xb} mod 4",
                fir_pattern(acc, g, x, "dup"))

    # dependency distance: fewer accumulators = the same register comes back sooner
    for na in (2, 4, 8, 16):
        add(f"dep_{na}", f"{na} accumulator pairs only (an accumulator is reused every {na} instructions)",
            fir_pattern(acc32[:na], g8, x16, "opsel", n_g=8 * 32 // na))

    # op_sel halves: only low-half / only high-half broadcasts
    ins = fir_pattern(acc32, g8, x16, "opsel")
    add("opsel_lo_only", "only broadcasts of the low half (op_sel_hi:[1,0,1])",
        [s.replace("op_sel:[0,1,0]", "op_sel_hi:[1,0,1]") for s in ins])
    add("opsel_hi_only", "only broadcasts of the high half (op_sel:[0,1,0])",
        [s.replace("op_sel_hi:[1,0,1]", "op_sel:[0,1,0]") for s in ins])

    # operand order: the broadcast x as src0 instead of src1
    def swap01(s):
        head, rest = s.split(" ", 1)
        ops = rest.split(", ")
        ops[1], ops[2] = ops[2], ops[1]
        tail = ops[3]
        tail = tail.replace("op_sel_hi:[1,0,1]", "op_sel_hi:[0,1,1]").replace("op_sel:[0,1,0]", "op_sel:[1,0,0]")
        ops[3] = tail
        return head + " " + ", ".join(ops)
    add("x_as_src0", "x pair as src0, taps as src1", [swap01(s) for s in ins])

    # reuse of the same tap (src0) in consecutive instructions is what the kernel does; alternate taps instead
    alt = []
    for o in range(32):
        for j in range(8):
            a = (o + 3 * j) % 32
            sel = "op_sel_hi:[1,0,1]" if (a & 1) == 0 else "op_sel:[0,1,0]"
            alt.append(f"v_pk_fma_f32 {pair(acc32[o])}, {pair(g8[j])}, {pair(x16[a >> 1])}, {pair(acc32[o])} {sel}")
    # (that order makes dependent chains of 8: kept as the worst case of back-to-back accumulation)
    add("chain8", "8 back-to-back FMAs into the same accumulator, then the next accumulator", alt)

    # filler instructions between packed FMAs: what issues beside them
    base = fir_pattern(acc32, g8, x16, "opsel")
    mixed = []
    for i, s in enumerate(base):
        mixed.append(s)
        if i % 8 == 7:
            mixed.append("s_nop 0")
    add("with_snop_1in8", "one s_nop 0 after every 8 packed FMAs", mixed)
    mixed = []
    for i, s in enumerate(base):
        mixed.append(s)
        if i % 8 == 7:
            mixed.append("s_add_u32 s20, s20, 1")
    add("with_salu_1in8", "one SALU add after every 8 packed FMAs", mixed)
    mixed = []
    for i, s in enumerate(base):
        mixed.append(s)
        if i % 16 == 15:
            mixed.append(f"v_mov_b32 v{200 + (i // 16) % 8}, v{16 + (i // 16) % 32}")
    add("with_vmov_1in16", "one v_mov_b32 after every 16 packed FMAs", mixed)
    # LDS reads in the stream (the row step reads 72 ds_read_b128 per 784 FMAs: ~1 per 11)
    mixed = []
    k = 0
    for i, s in enumerate(base):
        mixed.append(s)
        if i % 11 == 10:
            mixed.append(f"ds_read_b128 v[{192 + 4 * (k % 8)}:{195 + 4 * (k % 8)}], v15 offset:{16 * (k % 64)}")
            k += 1
            if k % 8 == 0:
                mixed.append("s_waitcnt lgkmcnt(4)")
    add("with_dsread_1in11", "one ds_read_b128 (broadcast address) after every 11 packed FMAs, counted waits", mixed)


def build_variants_b():
    """Set B: a steady stream of ONE instruction (all operands fixed; the accumulator chain is dependent, which
    costs nothing: dep_2 / chain8 of set A run at the full rate) over a grid of register numbers."""
    for A in (64, 66, 68, 70):
        for G in (192, 194, 196, 198):
            for X in (0, 2, 4, 6, 8, 10, 12, 14):
                add(f"fix_a{A}g{G}x{X}", f"one instruction repeated: acc v{A}, tap (src0) v{G}, x (src1) v{X}",
                    [f"v_pk_fma_f32 {pair(A)}, {pair(G)}, {pair(X)}, {pair(A)} op_sel_hi:[1,0,1]"] * 1024)


def build_variants_c():
    """Set C: sequence effects - which operand changes from one instruction to the next, and between which banks."""
    acc32 = [64 + 2 * o for o in range(32)]
    acc0 = [64 + 4 * o for o in range(32)]
    def seq(name, desc, accs, gfun, xfun, n=1024, sel="op_sel_hi:[1,0,1]"):
        ins = []
        for i in range(n):
            a = accs[i % len(accs)]
            ins.append(f"v_pk_fma_f32 {pair(a)}, {pair(gfun(i))}, {pair(xfun(i))}, {pair(a)} {sel}")
        add(name, desc, ins)
    for (x1, x2) in ((0, 4), (0, 2), (2, 6), (0, 8), (2, 10), (0, 6), (4, 6), (0, 0), (2, 2)):
        seq(f"xalt_{x1}_{x2}", f"src0 const v192, src1 alternates v{x1} / v{x2} every instruction, acc rotates over 32 consecutive pairs",
            acc32, lambda i: 192, lambda i, x1=x1, x2=x2: (x1, x2)[i & 1])
        seq(f"xalt2_{x1}_{x2}", f"same, src1 alternates every 2 instructions", acc32, lambda i: 192,
            lambda i, x1=x1, x2=x2: (x1, x2)[(i >> 1) & 1])
    for (g1, g2) in ((192, 196), (192, 194), (194, 198)):
        seq(f"galt_{g1}_{g2}", f"src1 const v0, src0 alternates v{g1} / v{g2} every instruction", acc32,
            lambda i, g1=g1, g2=g2: (g1, g2)[i & 1], lambda i: 0)
    # x walks over 16 pairs of one bank class, acc over pairs of one class (set A's slow / fast cases, one knob at a time)
    for xb in (0, 2):
        for step in (4, 8):
            seq(f"xwalk_b{xb}_s{step}", f"src1 walks v{xb} + {step} q (q = 0..{64 // step - 1}), new register every instruction",
                acc0, lambda i: 192, lambda i, xb=xb, step=step: xb + step * (i % (64 // step)))
    # the same walks with x in HIGH registers and the accumulators low: is it the register number or the role?
    for xb in (0, 2):
        seq(f"xwalk_hi_b{xb}", f"src1 walks v{128 + xb} + 4 q (q = 0..15), accumulators v0..v127 step 4", [4 * o for o in range(32)],
            lambda i: 192, lambda i, xb=xb: 128 + xb + 4 * (i % 16))
    # roles swapped: the varying broadcast operand as src0, the constant tap as src1 (r02's compiler-allocated ubench)
    for xb in (0, 2):
        for gb in (0, 2):
            seq(f"swap_x{xb}g{gb}", f"src0 walks v{xb} + 4 q (varying), src1 const v{192 + gb}", acc0,
                lambda i, xb=xb: xb + 4 * (i % 16), lambda i, gb=gb: 192 + gb, sel="op_sel_hi:[0,1,1]")
    seq("swap_nat", "src0 walks v16 + 2 q (varying, consecutive pairs), src1 const v48", acc32,
        lambda i: 16 + 2 * (i % 16), lambda i: 48, sel="op_sel_hi:[0,1,1]")
    seq("swap_nat_g50", "src0 walks v16 + 2 q, src1 const v50", acc32,
        lambda i: 16 + 2 * (i % 16), lambda i: 50, sel="op_sel_hi:[0,1,1]")


import sys
SETS = sys.argv[1] if len(sys.argv) > 1 else "a"
if "a" in SETS:
    build_variants()
if "b" in SETS:
    build_variants_b()
if "c" in SETS:
    build_variants_c()

HEADER = r"""// GENERATED by tools/gen_ubench_bank.py - do not edit.  See that file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef void (*kern_t)(float *, const float *, int, unsigned long long *);
"""


def kernel_text(name, ins):
    used = sorted(regs_of(ins) | {15})
    # v254 = lane * 4 (load offset), v255 = checksum; both outside every variant's maps
    assert max(used) < 250, (name, max(used))
    lines = []
    lines.append("v_mbcnt_lo_u32_b32 v254, -1, 0")
    lines.append("v_mbcnt_hi_u32_b32 v254, -1, v254")
    lines.append("v_lshlrev_b32 v254, 2, v254")
    lines.append("v_mov_b32 v15, 0")
    for i, r in enumerate(used):
        if r == 15:
            continue
        lines.append(f"global_load_dword v{r}, v254, %[in] offset:{(i * 260) % 3840}")
    lines.append("s_waitcnt vmcnt(0)")
    lines.append("s_mov_b32 s20, 0")
    lines.append("s_memtime %[c0]")
    lines.append("s_memrealtime %[r0]")
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_mov_b32 s21, %[iters]")
    lines.append("s_nop 4")
    lines.append(f"L_{name}_%=:")
    lines.extend(ins)
    lines.append("s_sub_u32 s21, s21, 1")
    lines.append("s_cmp_lg_u32 s21, 0")
    lines.append(f"s_cbranch_scc1 L_{name}_%=")
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_memtime %[c1]")
    lines.append("s_memrealtime %[r1]")
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("v_mov_b32 v255, 0")
    for r in used:
        lines.append(f"v_add_f32 v255, v255, v{r}")
    lines.append("v_mov_b32 %[res], v255")
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{r}"' for r in used + [254, 255])
    return f"""
__global__ __launch_bounds__(256) void k_{name}(float *out, const float *in, int iters, unsigned long long *stamps) {{
    unsigned long long c0, c1, r0, r1;
    float res;
    asm volatile(
{body}
        : [c0] "=&s"(c0), [c1] "=&s"(c1), [r0] "=&s"(r0), [r1] "=&s"(r1), [res] "=v"(res)
        : [in] "s"(in), [iters] "s"(iters)
        : "memory", "s20", "s21", "scc", {clob});
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (threadIdx.x == 0) {{
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }}
}}
"""


MAIN = r"""
struct Variant { const char *name; const char *desc; kern_t fn; int n_ins; int flop_per_ins; };

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 0.3;     // back-to-back launches per line
    const char *only = argc > 2 ? argv[2] : nullptr;
    float *out, *in;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 1 << 24);
    (void)hipMalloc(&in, 8192);
    (void)hipMalloc(&stamps, 2 * 2048 * sizeof(unsigned long long));
    static float hbuf[2048];
    const int iters = 200;
    printf("# gfx950 VALU issue microbenchmark with explicit VGPR numbers (tools/gen_ubench_bank.py); %g s per line\n", seconds);
    printf("# variant operands waves/SIMD  cycles/instr/wave  cycles/instr/SIMD  clock-GHz  TFLOP/s   | description\n");
    for (size_t v = 0; v < sizeof(variants) / sizeof(variants[0]); ++v) {
        const Variant &V = variants[v];
        if (only && !strstr(V.name, only)) continue;
        for (int random = 0; random <= 1; ++random) {
            unsigned s = 12345u;
            for (int i = 0; i < 2048; ++i) {
                s = s * 1664525u + 1013904223u;
                hbuf[i] = random ? ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.01f : 0.0f;
            }
            (void)hipMemcpy(in, hbuf, 8192, hipMemcpyHostToDevice);
            for (int wps = 1; wps <= 2; ++wps) {
                const int blocks = 256 * wps;
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                hipLaunchKernelGGL(V.fn, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
                (void)hipDeviceSynchronize();
                float ms1 = 0.f;
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(V.fn, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
                (void)hipEventRecord(e1);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms1, e0, e1);
                int reps = (int)(seconds * 1e3 / (ms1 > 0.01f ? ms1 : 0.01f));
                if (reps < 3) reps = 3;
                (void)hipEventRecord(e0);
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(V.fn, dim3(blocks), dim3(256), 0, 0, out, in, iters, stamps);
                (void)hipEventRecord(e1);
                (void)hipDeviceSynchronize();
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                ms /= reps;
                std::vector<unsigned long long> st(2 * blocks);
                (void)hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::vector<double> ghz(blocks), cyc(blocks);
                for (int b = 0; b < blocks; ++b) {
                    ghz[b] = st[2 * b + 1] ? (double)st[2 * b] / (double)st[2 * b + 1] * 0.1 : 0.0;
                    cyc[b] = (double)st[2 * b] / ((double)V.n_ins * iters);
                }
                std::sort(ghz.begin(), ghz.end());
                std::sort(cyc.begin(), cyc.end());
                const double flops = (double)V.n_ins * iters * V.flop_per_ins * 4.0 * blocks;
                printf("%-18s %-6s %d  %7.3f  %7.3f  %.3f  %6.1f   | %s\n", V.name, random ? "random" : "zero", wps,
                       cyc[blocks / 2], cyc[blocks / 2] / wps, ghz[blocks / 2], flops / (ms * 1e-3) / 1e12, V.desc);
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            }
        }
        fflush(stdout);
    }
    hipError_t e = hipGetLastError();
    printf("# %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
"""


def main():
    out = [HEADER]
    for name, desc, ins, fl in VARIANTS:
        out.append(kernel_text(name, ins))
    out.append("static const Variant variants[] = {")
    for name, desc, ins, fl in VARIANTS:
        n_vec = sum(1 for s in ins if s.startswith("v_pk") or s.startswith("v_fma"))
        out.append(f'    {{"{name}", "{desc}", k_{name}, {n_vec}, {fl}}},')
    out.append("};")
    out.append(MAIN.replace("struct Variant {", "struct Variant_unused {", 1) if False else "")
    text = "\n".join(out)
    # the Variant struct must precede the table
    text = text.replace("static const Variant variants[] = {",
                        "struct Variant { const char *name; const char *desc; kern_t fn; int n_ins; int flop_per_ins; };\n"
                        "static const Variant variants[] = {")
    text += MAIN.replace("struct Variant { const char *name; const char *desc; kern_t fn; int n_ins; int flop_per_ins; };\n", "")
    with open(os.path.join(HERE, "ubench_bank_gen.hip"), "w") as f:
        f.write(text)
    print(f"{len(VARIANTS)} variants -> tools/ubench_bank_gen.hip")


if __name__ == "__main__":
    main()
