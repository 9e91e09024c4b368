#!/bin/bash
# Variant of the HIP library that differs from the in-tree build only in ONE source file (default attn.hip):
#   tools/build_attn_variant.sh NAME "-DMU_FLAG=1 ..." [file]     -> gpurun_variants/libmu_NAME.so  (tools/ab_bench.py)
# The other objects are the in-tree ones (run `make -C maskunet_amd/csrc` first).  Debug aid.
set -e
NAME=$1; EXTRA="$2"; F=${3:-attn}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/maskunet_amd/csrc
OUT=$ROOT/gpurun_variants
mkdir -p $OUT
FL=""; { [ $F = attn ] || [ $F = conv ]; } && FL="-fno-slp-vectorize"      # as the Makefile builds these two (FLAGS_attn / FLAGS_conv)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-pass-failed $FL $EXTRA -c $SRC/$F.hip -o $OUT/${F}_$NAME.o
OBJS=""
for f in elementwise norm conv attn attn_wide loss probe version; do
  if [ $f = $F ]; then OBJS="$OBJS $OUT/${F}_$NAME.o"; else OBJS="$OBJS $SRC/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libmu_$NAME.so
rm -f $OUT/${F}_$NAME.o
echo built $OUT/libmu_$NAME.so
