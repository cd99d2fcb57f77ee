#!/bin/bash
# Runs ON THE GPU BOX (gpurun): collects every round-3 measurement DESIGN.md / README.md quote into gpurun_out/r03p/.
# tools/publish_profiles.py r03p r03_ then copies the summaries into profiles/ (tracked).
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles_r03.sh'
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03p
mkdir -p $O
B="python3 bench.py"
C=$PWD/binaural-audio-synthesis_amd/csrc
# 1. the driver's command, and a long run
$B > $O/bench_default.json 2> $O/bench_default.err
$B --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_300.json 2>> $O/bench_default.err
echo "[1] bench done"
# 2. rocprofv3 kernel stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err
echo "[2] kernel stats done"
# 3. PMC passes (separate runs): HBM traffic of the FIR kernel, SQ counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/pmc_SQ -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > /dev/null 2> $O/pmc_SQ.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_SQ2 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-self-check > /dev/null 2> $O/pmc_SQ2.err
echo "[3] pmc done"
# 4. unfused path for comparison (interp2d + hd kernel), same box
$B --unfused --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_unfused -o bench -- python3 bench.py --unfused --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/prof_unfused.err
# 5. single source (BASELINE configs 2 / 3)
$B --sources 1 --steps 300 --warmup 10 --no-cpu-baseline > $O/single_source.json 2> $O/single.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o bench -- python3 bench.py --sources 1 --steps 300 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/single.err
python3 tools/single_source_latency.py > $O/single_source_latency.txt 2>&1
echo "[5] single done"
# 6. streaming: config 5 - the WHOLE hour (659 blocks of 262 144 samples at 48 kHz), kernel stats of 40 blocks, the
#    collective path on one rank, host time of real-time sized blocks
$B --mode stream --sources 1024 --fs 48000 --steps 659 --warmup 3 > $O/stream_hour.json 2> $O/stream.err
$B --mode stream --sources 1024 --fs 48000 --steps 659 --warmup 3 --regen > $O/stream_hour_regen.json 2>> $O/stream.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o bench -- python3 bench.py --mode stream --sources 1024 --fs 48000 --steps 40 --warmup 3 > /dev/null 2>> $O/stream.err
$B --mode stream --sources 1024 --fs 48000 --steps 100 --warmup 3 --force-pg > $O/stream_forcepg.json 2>> $O/stream.err
python3 tools/stream_host_time.py 256 512 2>/dev/null > $O/stream_host_time.txt
python3 tools/stream_host_time.py 256 32768 2>/dev/null >> $O/stream_host_time.txt
echo "[6] stream done"
# 7. per-rank shares of the strong-scaling split on one GPU (what a rank of N = 8 / 4 / 2 renders), graph-replayed
#    and plain, and the collective path on one rank (--force-pg: RCCL communicator of size 1)
for n in 32 64 128 256; do
  $B --sources $n --steps 200 --warmup 10 --no-cpu-baseline --no-self-check --graph on > $O/share_${n}_graph.json 2>> $O/share.err
  $B --sources $n --steps 200 --warmup 10 --no-cpu-baseline --no-self-check --graph off > $O/share_${n}_plain.json 2>> $O/share.err
  $B --sources $n --steps 200 --warmup 10 --no-cpu-baseline --no-self-check --force-pg > $O/share_${n}_forcepg.json 2>> $O/share.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_share32 -o bench -- python3 bench.py --sources 32 --steps 200 --warmup 10 --no-cpu-baseline --no-self-check > /dev/null 2>> $O/share.err
echo "[7] shares done"
# 8. the two fused kernels in one process, interleaved rounds: two workgroups per CU (every wave stages and filters;
#    make -C csrc nosplit) against the split-role kernel the shipped library picks for these scenes
( cd $C && [ -f libbas_hip_nosplit.so ] || echo "missing libbas_hip_nosplit.so (make nosplit)"
  for n in 256 128 64 32; do
    echo "== $n sources"; python3 ../../tools/ab_fir.py --sources $n --rounds 5 --reps 10 libbas_hip_nosplit.so libbas_hip.so 2>&1 | tail -2
  done > $O/ab_split_roles.txt 2>&1 )
# 9. where the waves of the split-role kernel spend their time (diagnostic build), and what a wave that has its SIMD to
#    itself pays for the non-VALU instructions of the row step (tools/ubench_lone_wave.hip, variants of tools/gen_fir_asm.py)
python3 tools/stamps_fs.py 2>/dev/null > $O/stamps_fs_256.txt
( cd tools && for v in generic straight nobranch nowait notaps nox noalign valuonly; do
    [ -x ./ubench_lone_$v ] && ./ubench_lone_$v 400 $v; done
  [ -x ./ubench_lone_generic ] && ./ubench_lone_generic 400 generic-unit 0
  [ -x ./ubench_lone_straight ] && ./ubench_lone_straight 400 straight-unit 0 ) > $O/ubench_lone_wave.txt 2>&1
# 10. two ranks on one device (rehearsal of the multi-rank bench path under gloo)
$B --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
echo collected
