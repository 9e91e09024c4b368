#!/bin/bash
# Round-6 evidence session: everything under profiles/r06_* (except the A/B files, which name their own sessions) comes from ONE gpurun
# call of this script on ONE box.   bash tools/session_r06.sh TAG    (from the repo root on the GPU box; outputs in gpurun_out/<TAG>*)
TAG=${1:-r06z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/gpu_session.sh $TAG tests bench
# kernel trace kept for the per-launch durations of the roofline kernel
export TMPDIR=/tmp
( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --steps 8 --warmup 2 > $ROOT/gpurun_out/${TAG}_prof.log 2>&1 )
f=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1)
python tools/prof_summary.py $f 10 40 > gpurun_out/${TAG}_prof_summary.txt; cp $f gpurun_out/${TAG}_kernel_stats.csv
python tools/dkv_launches.py gpurun_out/${TAG}_prof gpurun_out/${TAG}_bench.json > gpurun_out/${TAG}_dkv_launches.txt 2>&1
bash tools/gpu_session.sh $TAG pmc
( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}x_prof -o a -- python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --no-clock-probe --no-fp32x-line --no-host-inclusive --dtype fp32x --steps 8 --warmup 2 > $ROOT/gpurun_out/${TAG}x_prof.log 2>&1 )
f=$(find gpurun_out/${TAG}x_prof -name '*kernel_stats.csv' | head -1)
python tools/prof_summary.py $f 10 40 > gpurun_out/${TAG}x_prof_summary.txt; cp $f gpurun_out/${TAG}x_kernel_stats.csv
MU_SESSION_FLAGS="--dtype fp32x" bash tools/gpu_session.sh ${TAG}x bench pmc
bash tools/lds_conflicts.sh ${TAG} > gpurun_out/${TAG}_lds_conflicts_fp16.md 2>&1
bash tools/lds_conflicts.sh ${TAG}x --dtype fp32x > gpurun_out/${TAG}_lds_conflicts_fp32x.md 2>&1
bash tools/bench_configs.sh $TAG > /dev/null
python tools/bench_layers.py 64 > gpurun_out/${TAG}_conv_layers.md 2>&1
python tools/bench_layers.py 64 --fp32x > gpurun_out/${TAG}_conv_layers_fp32x.md 2>&1
bash tools/dkv_traffic.sh $TAG b64_c150_hw128_fp16
bash tools/dkv_traffic.sh $TAG b128_c133_hw128_fp16 --c-out 133 --batch 128
bash tools/dkv_traffic.sh $TAG b64_c19_hw128_fp16_3head --three-head --c-out 19 --batch 64
bash tools/dkv_traffic.sh $TAG b32_c133_hw256_fp16 --hw 256 --c-out 133 --batch 32
bash tools/dkv_traffic.sh $TAG b64_c150_hw128_fp32 --dtype fp32
bash tools/dkv_traffic.sh $TAG b64_c150_hw128_fp32x --dtype fp32x
rm -rf gpurun_out/${TAG}_traf_* gpurun_out/${TAG}*_pmc_sq gpurun_out/${TAG}*_pmc_fetch gpurun_out/${TAG}*_pmc_write gpurun_out/${TAG}*_lds gpurun_out/${TAG}*_prof
echo "session $TAG done"
