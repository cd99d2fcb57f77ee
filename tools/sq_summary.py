#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc run (counter_collection.csv): mean counters per dispatch and, where the
SQ counters are present, VALU busy, implied clock and instructions per wave.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \\
        --output-format csv -d gpurun_out/pmc -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/sq_summary.py gpurun_out/pmc/sq_counter_collection.csv [kernel-name-substring]

SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md): x4 for cycles.  The implied clock
assumes the waves live for the whole dispatch (true for the persistent FIR grid, not for short-wave kernels)."""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if want not in name:
            continue
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]),
                      int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
    for name, c in per.items():
        grid, wg, dur_ns, vgpr, lds = meta[name]
        waves = grid / 64
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        print(f"{name[:90]}\n  grid {grid} x wg {wg}, {waves:.0f} waves, {vgpr} VGPR, {lds} B LDS, last dispatch {dur_ns / 1e3:.1f} us")
        for k, v in sorted(mean.items()):
            print(f"  {k:26s} {v:14.5g}   per wave {v / waves:12.5g}")
        if "SQ_WAVE_CYCLES" in mean and "SQ_ACTIVE_INST_VALU" in mean:
            life = 4 * mean["SQ_WAVE_CYCLES"] / waves
            simds = 1024.0
            busy = 4 * mean["SQ_ACTIVE_INST_VALU"] / simds
            print(f"  wave lifetime {life:.4g} cycles -> clock {life / dur_ns:.2f} GHz if waves span the dispatch; "
                  f"VALU busy per SIMD {busy:.4g} cycles = {100 * busy / life:.1f} % of a wave lifetime")


if __name__ == "__main__":
    main()
