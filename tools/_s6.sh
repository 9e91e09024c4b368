cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider > gpurun_out/r06f_pytest.log 2>&1; grep -E "passed|failed|FAILED|parity failures" gpurun_out/r06f_pytest.log | cut -c1-600 | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06f_bench.json").read().strip().splitlines()[-1])
print("fp16", r["value"], r["ms_per_step"], "| parity-grade", r["parity_grade_path"]["value"], r["parity_grade_path"]["ms_per_step"], r["parity_grade_path"].get("parity_gate", {}).get("observed"))
PY
