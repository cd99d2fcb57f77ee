#!/bin/bash
# quick look at bench.py --overlap-plans (the plans of step i+1 beside the FIR of step i) on one GPU
set -u
O=$PWD/gpurun_out/ovl; mkdir -p $O
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-self-check --no-traffic"
for n in 32 27 64 256; do
  for m in off on; do
    $B --sources $n --graph off --overlap-plans $m > $O/plain_${n}_$m.json 2>> $O/err.txt
    $B --sources $n --graph on --overlap-plans $m > $O/graph_${n}_$m.json 2>> $O/err.txt
    BAS_BENCH_CHECK=1 $B --sources $n --force-pg --overlap-plans $m > $O/forcepg_${n}_$m.json 2>> $O/err.txt
  done
done
python3 - <<'PY'
import json,glob,os
O=os.path.join(os.getcwd(),"gpurun_out/ovl")
for n in (27,32,64,256):
    row=[f"{n:4d} sources"]
    for kind in ("plain","graph","forcepg"):
        for m in ("off","on"):
            try:
                j=json.load(open(f"{O}/{kind}_{n}_{m}.json")); row.append(f"{kind}/{m} {j['ms_per_step']*1e3:7.1f} (FIR {j['roofline']['kernel_ms']*1e3:6.1f})")
            except Exception as e: row.append(f"{kind}/{m} -")
    print(" | ".join(row))
PY
grep -c "check:" $O/err.txt; tail -5 $O/err.txt
