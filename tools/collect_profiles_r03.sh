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
# 8. ablations, one process each, interleaved rounds: no chunk-IR evaluation (floor of the FIR part), the previous
#    round's evaluation scheme (21 IRs per pass), packed runs aligned to 8 bytes
( cd $C && for lib in libbas_noeval.so libbas_prev.so; do [ -f $lib ] || echo "missing $lib"; done
  python3 ../../tools/ab_fir.py --rounds 7 --reps 20 --no-check libbas_hip.so libbas_noeval.so > $O/ab_noeval.txt 2>&1
  python3 ../../tools/ab_fir.py --rounds 7 --reps 20 libbas_prev.so libbas_hip_cppstep.so libbas_hip.so > $O/ab_ir_sharing.txt 2>&1 )
# 9. phase stamps of the fused kernel (diagnostic build)
python3 tools/stamps_fz.py 256 2>/dev/null > $O/stamps_fz_256.txt
# 10. two ranks on one device (rehearsal of the multi-rank bench path under gloo)
$B --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
./tools/ubench_bank 0.3 > $O/ubench_bank.txt 2>&1
echo collected
