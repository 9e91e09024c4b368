cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tests/dkv_traffic.sh r03 b64_c150_hw128_fp16 2>&1 | tail -3
bash tests/dkv_traffic.sh r03 b128_c133_hw128_fp16 --c-out 133 --batch 128 2>&1 | tail -3
bash tests/dkv_traffic.sh r03 b64_c19_hw128_fp16_3head --three-head --c-out 19 --batch 64 2>&1 | tail -3
bash tests/dkv_traffic.sh r03 b32_c133_hw256_fp16 --hw 256 --c-out 133 --batch 32 2>&1 | tail -3
bash tests/dkv_traffic.sh r03 b64_c150_hw128_fp32 --dtype fp32 2>&1 | tail -3
bash tests/bench_configs.sh r03 2>&1 | tail -12
