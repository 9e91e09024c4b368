#!/bin/bash
# Build a variant of the HIP library for in-process A/B timing: tools/build_variant.sh NAME "-DMU_FLAG=1 ..."
# -> gpurun_variants/libmu_NAME.so (used by tools/ab_bench.py).  Debug aid, not part of the product build.
set -e
NAME=$1; shift
EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${MU_SRC:-$ROOT/maskunet_amd/csrc}      # MU_SRC=<dir> builds another checkout's sources (e.g. a git worktree of HEAD)
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$NAME
for f in elementwise norm conv attn attn_wide loss probe version; do
  FL=""; [ $f = attn ] && FL="-fno-slp-vectorize"      # as the Makefile builds attn.hip (FLAGS_attn)
  [ $f = conv ] && [ -z "$MU_CONV_SLP" ] && FL="-fno-slp-vectorize"      # FLAGS_conv (MU_CONV_SLP=1: the former build)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-pass-failed $FL $EXTRA -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj_$NAME/*.o -o $OUT/libmu_$NAME.so
rm -rf $OUT/obj_$NAME
echo built $OUT/libmu_$NAME.so
