#!/usr/bin/env python3
"""Insert `.p2align 3` (the assembler pads code with s_nop 0) in front of every run of >= MIN_RUN consecutive 8-byte
packed-f32 VALU instructions of a gfx950 assembly file: a 64-bit instruction at an address = 4 mod 8 costs one more
issue cycle (profiles/r03_ubench_bank_placement.txt)."""
import re, sys
PK = re.compile(r"^\s+(v_pk_fma_f32|v_pk_add_f32|v_pk_mul_f32)\s")
def main(src, dst, min_run=3):
    lines = open(src).read().split("\n")
    out = []; i = 0; n_pad = 0
    while i < len(lines):
        if PK.match(lines[i]):
            j = i
            while j < len(lines) and PK.match(lines[j]): j += 1
            if j - i >= min_run:
                out.append("\t.p2align 3"); n_pad += 1
            out.extend(lines[i:j]); i = j
        else:
            out.append(lines[i]); i += 1
    open(dst, "w").write("\n".join(out))
    print(f"{src}: {n_pad} alignment points", file=sys.stderr)
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
