#!/bin/bash
# Round-5 evidence session: everything under profiles/r05_* comes from ONE gpurun call of this script on ONE box (VERDICT r4 #5c).
#     bash tools/session_r05.sh TAG        (from the repo root on the GPU box; outputs in gpurun_out/<TAG>*)
TAG=${1:-r05z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/gpu_session.sh $TAG tests bench prof pmc
MU_SESSION_FLAGS="--dtype fp32x" bash tools/gpu_session.sh ${TAG}x bench prof pmc
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
for t in $TAG ${TAG}x; do f=$(find gpurun_out/${t}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${t}_kernel_stats.csv; done
rm -rf gpurun_out/${TAG}_traf_* gpurun_out/${TAG}*_pmc_sq gpurun_out/${TAG}*_pmc_fetch gpurun_out/${TAG}*_pmc_write gpurun_out/${TAG}*_lds gpurun_out/${TAG}*_prof
echo "session $TAG done"
