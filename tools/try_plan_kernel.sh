#!/bin/bash
# kernel times of the small-scene chains (plan kernel with the ring search from LDS)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/plank; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "interp or a3 or traj or plan or angles or stream" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for n in 1 32 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$n -o b -- python3 bench.py --no-traffic --sources $n --steps 200 --warmup 10 --no-cpu-baseline --no-self-check > $O/b$n.json 2> $O/b$n.err
  echo "== $n sources: $(python3 -c "import json;print(json.load(open('$O/b$n.json'))['ms_per_step']*1e3)") us/step (profiled)"
  grep "plan_kernel\|fs_kernel\|fq_kernel\|reduce\|scale_kernel" $(find $O/p$n -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-60,100-
done
python3 bench.py --sources 1 --steps 300 --warmup 10 --no-cpu-baseline --no-traffic | python3 -c "import json,sys;print('single source us/step', json.loads(sys.stdin.read())['ms_per_step']*1e3)"
python3 bench.py --sources 32 --steps 300 --warmup 10 --no-cpu-baseline --no-traffic | python3 -c "import json,sys;print('32 sources us/step', json.loads(sys.stdin.read())['ms_per_step']*1e3)"
python3 tools/stream_host_time.py 256 512 2>/dev/null
