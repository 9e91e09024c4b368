#!/bin/bash
# fp32x A/B session (debug aid): kernel + module parity in the split-bf16 mode, the bench line and the rocprof kernel split.
#     bash tools/session_fp32x.sh TAG      (from the repo root on the GPU box)
TAG=${1:-fx}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "fp32x or conv" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log
timeout 600 python bench.py --dtype fp32x --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_fp32x.json 2> $OUT/${TAG}_bench_fp32x.err
echo "bench rc=$?"; cut -c1-400 $OUT/${TAG}_bench_fp32x.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_fp32x -o a -- python3 $ROOT/bench.py --dtype fp32x --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 6 --warmup 2 > $OUT/${TAG}_prof_fp32x.log 2>&1
echo "prof rc=$?"
cd $ROOT
f=$(find $OUT/${TAG}_prof_fp32x -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python tools/prof_summary.py $f 8 30 > $OUT/${TAG}_prof_fp32x_summary.txt && head -40 $OUT/${TAG}_prof_fp32x_summary.txt
rm -rf $OUT/${TAG}_prof_fp32x
