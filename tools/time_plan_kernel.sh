#!/bin/bash
# the plan kernel (a3 + read plans, one launch) for 1 / 32 / 256 sources x 10 s under rocprofv3, and the step around it
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/plank; mkdir -p $O
for n in 1 32 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$n -o b -- python3 bench.py --no-traffic --graph off --sources $n --steps 200 --warmup 10 --no-cpu-baseline --no-self-check > $O/b$n.json 2> $O/b$n.err
  python3 - <<PY
import csv, json
j = json.load(open("$O/b$n.json"))
print(f"== $n sources: {j['ms_per_step'] * 1e3:.1f} us/step (profiled)")
for r in csv.DictReader(open("$O/p$n/b_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("plan_kernel", "fs_kernel", "fq_kernel", "reduce", "scale_kernel")):
        print(f'   {r["Name"][:45]:45s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
done
