#!/bin/bash
# bench.py on every BASELINE.json configuration shape that fits one GPU (+ opt-in variants); one JSON line each -> gpurun_out/<TAG>_configs.jsonl
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_configs.jsonl
: > $OUT
cd $ROOT
run() { echo "# bench.py $*" >> $OUT; timeout 900 python bench.py --no-cpu-baseline --no-fp32x-line --no-host-inclusive --steps 10 --warmup 3 "$@" 2>>$ROOT/gpurun_out/${TAG}_configs.err | grep '^{' >> $OUT; }
run                                               # configs[1]: ADE20K semantic, B=64, fp16
run --c-out 133 --batch 128                       # configs[2]: COCO panoptic shape, B=128
run --three-head --c-out 19 --batch 64            # configs[3] per-GPU shape: Cityscapes instance, 3-head, B=64/GPU
run --hw 256 --c-out 133 --batch 32               # configs[4] per-GPU shape: COCO semantic 256x256, B=32/GPU
run --graph                                       # configs[1] replayed as one HIP graph
run --optimizer --warmup 8                       # configs[1] + fused AdamW inside the step (one-time allocator / pinned-ring costs land in steps 4-7: longer warm-up)
run --dtype fp32 --steps 4 --warmup 2             # fp32 parity path (exact-fp32 MFMA)
run --dtype fp32x --steps 6 --warmup 2            # fp32 storage, split-bf16 matrix products (set_float32_matmul_precision("high"))
run --batch 16
# context only: the reference's own batch range (1-14, ade_semantic.py:18) is launch-bound on this path
run --batch 8
run --batch 8 --graph
run --batch 16 --graph
# the other configuration shapes in the fp32x mode
run --dtype fp32x --c-out 133 --batch 128 --steps 4 --warmup 2
run --dtype fp32x --three-head --c-out 19 --batch 64 --steps 6 --warmup 2
run --dtype fp32x --hw 256 --c-out 133 --batch 32 --steps 4 --warmup 2
cat $OUT | cut -c1-260
