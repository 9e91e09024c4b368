cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -k "fp32x" --timeout 900 -p no:cacheprovider > gpurun_out/r06e_fp32x.log 2>&1; grep -E "fp32x unet|passed|failed|FAILED" gpurun_out/r06e_fp32x.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-host-inclusive --dtype fp32x --steps 10 --warmup 3 > gpurun_out/r06e_bench_fp32x.json 2> gpurun_out/r06e_bench_fp32x.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06e_bench_fp32x.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], json.dumps(r.get("parity_gate"))[:600])
print(json.dumps(r.get("roofline"))[:900])
PY
