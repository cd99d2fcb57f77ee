set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03
mkdir -p $O
cd binaural-audio-synthesis_amd/csrc
python ../../tools/ab_fir.py --rounds 5 --reps 20 --zero-x libbas_hip.so libbas_hip_al.so > $O/ab_align_zero.txt 2>&1
python ../../tools/ab_fir.py --rounds 5 --reps 20 libbas_hip_al.so libbas_hip.so > $O/ab_align_swapped.txt 2>&1
cd ../..
for lib in libbas_hip libbas_hip_al; do
  LIBARG="--lib $PWD/binaural-audio-synthesis_amd/csrc/$lib.so"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_$lib -o pmc -- python3 bench.py $LIBARG --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmc_$lib.err
  python3 tools/sq_summary.py $O/pmc_$lib/pmc_counter_collection.csv fz_kernel > $O/sq_$lib.txt 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmcg_$lib -o pmc -- python3 bench.py $LIBARG --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmcg_$lib.err
  python3 tools/sq_summary.py $O/pmcg_$lib/pmc_counter_collection.csv fz_kernel > $O/grbm_$lib.txt 2>&1
  rm -rf $O/pmc_$lib $O/pmcg_$lib
done
cat $O/ab_align_zero.txt $O/ab_align_swapped.txt $O/sq_*.txt $O/grbm_*.txt
