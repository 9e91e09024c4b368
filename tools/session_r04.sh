bash tools/gpu_session.sh r04g tests bench prof pmc
bash tools/bench_configs.sh r04g
python tools/bench_layers.py 64 > gpurun_out/r04g_conv_layers.md 2>&1
bash tools/dkv_traffic.sh r04g b64_c150_hw128_fp16
bash tools/dkv_traffic.sh r04g b128_c133_hw128_fp16 --c-out 133 --batch 128
bash tools/dkv_traffic.sh r04g b64_c19_hw128_fp16_3head --three-head --c-out 19 --batch 64
bash tools/dkv_traffic.sh r04g b32_c133_hw256_fp16 --hw 256 --c-out 133 --batch 32
bash tools/dkv_traffic.sh r04g b64_c150_hw128_fp32 --dtype fp32
bash tools/dkv_traffic.sh r04g b64_c150_hw128_fp32x --dtype fp32x
rm -rf gpurun_out/r04g_traf_* gpurun_out/r04g_pmc_sq gpurun_out/r04g_pmc_fetch gpurun_out/r04g_pmc_write
