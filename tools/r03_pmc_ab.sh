#!/bin/bash
# Runs ON THE GPU BOX: A/B of builds of the library (same ABI) - FIR kernel time in one process (interleaved rounds,
# noise and zero input) and the SQ counters of the fused kernel for each build.
#   gpurun -- 'bash tools/r03_pmc_ab.sh <tag> libA.so libB.so ...'     (files of binaural-audio-synthesis_amd/csrc)
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=$PWD/gpurun_out/r03
mkdir -p $O
C=$PWD/binaural-audio-synthesis_amd/csrc
( cd $C && python ../../tools/ab_fir.py --rounds 7 --reps 20 "$@" > $O/ab_${TAG}.txt 2>&1
  python ../../tools/ab_fir.py --rounds 5 --reps 20 --zero-x "$@" > $O/ab_${TAG}_zero.txt 2>&1 )
for lib in "$@"; do
  name=${lib%.so}
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_$name -o pmc -- python3 bench.py --lib $C/$lib --steps 20 --warmup 5 --no-cpu-baseline --no-self-check > /dev/null 2> $O/pmc_$name.err
  python3 tools/sq_summary.py $O/pmc_$name/pmc_counter_collection.csv fz_kernel > $O/sq_${TAG}_$name.txt 2>&1
  rm -rf $O/pmc_$name
done
grep -v amdgpu.ids $O/ab_${TAG}.txt $O/ab_${TAG}_zero.txt; cat $O/sq_${TAG}_*.txt
