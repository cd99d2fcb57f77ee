#!/bin/bash
# Builds A/B partners of the shipped library that differ ONLY in the generated assembly of the split-role kernel's unit block:
#   tools/build_fir_variants.sh NAME "generator flags" [NAME "flags" ...]
# -> binaural-audio-synthesis_amd/csrc/libab_NAME.so (bas_fused_split.hip recompiled against the variant, the shipped objects
#    for everything else) and tools/ubench_unit_NAME (tools/ubench_unit_block.hip on the same variant).  Both git-ignored.
set -e
cd "$(dirname "$0")/.."
C=binaural-audio-synthesis_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
make -s -C $C libbas_hip.so
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  inc=$PWD/$C/ab_$name.inc
  python3 tools/gen_fir_asm.py $flags --out=$inc > /dev/null
  ( $HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBAS_FIR_ASM_INC="\"$inc\"" -c $C/bas_fused_split.hip -o $C/ab_$name.o &&
    $HIPCC -shared -fPIC --offload-arch=gfx950 $C/bas_abi.o $C/bas_interp.o $C/bas_render.o $C/bas_fused.o $C/ab_$name.o $C/bas_fused_quad.o $C/bas_stream.o -o $C/libab_$name.so ) &
  $HIPCC -O3 -w --offload-arch=gfx950 -DFIR_INC="\"$inc\"" tools/ubench_unit_block.hip -o tools/ubench_unit_$name &
  wait
  echo "built $name ($flags)"
done
