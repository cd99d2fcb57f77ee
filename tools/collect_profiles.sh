#!/bin/bash
# Runs ON THE GPU BOX (gpurun): collects every measurement DESIGN.md / README.md quote into gpurun_out/r02p/.
# tools/publish_profiles.py then filters the large CSVs and copies the summaries into profiles/ (tracked).
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'
set -u
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
B="python3 bench.py"
# 1. the driver's command, twice (box spread), and a long run
$B > $O/bench_default.json 2> $O/bench_default.err
$B --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_300.json 2>> $O/bench_default.err
# 2. rocprofv3 kernel stats of the same command (>= 200 steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err
# 3. PMC passes (separate runs): HBM traffic of the FIR kernel, SQ counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/pmc_SQ -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_SQ.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_SQ2 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_SQ2.err
# 4. unfused path for comparison (interp2d + hd kernel), same box
$B --unfused --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_unfused -o bench -- python3 bench.py --unfused --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/prof_unfused.err
# 5. single source (BASELINE configs 2 / 3)
$B --sources 1 --steps 300 --warmup 10 --no-cpu-baseline > $O/single_source.json 2> $O/single.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o bench -- python3 bench.py --sources 1 --steps 300 --warmup 10 --no-cpu-baseline > /dev/null 2>> $O/single.err
BAS_FZ_NW=4 $B --lib $PWD/binaural-audio-synthesis_amd/csrc/libbas_hip_diag.so --sources 1 --steps 300 --warmup 10 --no-cpu-baseline > $O/single_source_tile8192.json 2>> $O/single.err
python3 tools/single_source_latency.py > $O/single_source_latency.txt 2>&1
# 6. streaming: config 5 shape, and host time of real-time sized blocks
$B --mode stream --sources 1024 --fs 48000 --steps 20 --warmup 3 > $O/stream_1024src_48k.json 2> $O/stream.err
$B --mode stream --sources 1024 --fs 48000 --steps 20 --warmup 3 --regen > $O/stream_1024src_48k_regen.json 2>> $O/stream.err
python3 tools/stream_host_time.py 256 512 2>/dev/null > $O/stream_host_time.txt
python3 tools/stream_host_time.py 256 32768 2>/dev/null >> $O/stream_host_time.txt
# 7. the ceiling evidence: bare packed-FMA stream (zero / random operands, 1 / 2 waves per SIMD, in-kernel clock)
./tools/ubench_fir_pattern 1.0 > $O/ubench_fir_pattern.txt 2>&1
./tools/ubench_fma_forms 0.5 > $O/ubench_fma_forms.txt 2>&1
./tools/ubench_fir_steps > $O/ubench_fir_steps.txt 2>&1
python3 tools/warmup_series.py 2>/dev/null > $O/warmup_series.txt
# 8. phase stamps of the fused kernel (diagnostic build)
python3 tools/stamps_fz.py 256 2>/dev/null > $O/stamps_fz_256.txt
# 9. two ranks on one device (rehearsal of the multi-rank bench path)
$B --gpus 2 --steps 20 --warmup 3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
echo collected
