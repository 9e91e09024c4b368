#!/usr/bin/env python3
"""Copy the outputs of ONE evidence session (tools/session_r06.sh TAG -> gpurun_out/TAG*) into profiles/r06_*:
    python tools/collect_r06.py TAG
Every copied text file gets a first line naming the session, so a reader can tell which build / box a number comes from."""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
NOTE = (f"session {tag}: one gpurun call of tools/session_r06.sh on one MI355X box, the round's final build; fp32x files from the same "
        f"call (tag {tag}x)")


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return lines[-1]


def md(src, dst, title=None):
    body = [l for l in open(os.path.join(G, src)) if "amdgpu.ids" not in l and not l.startswith("rc=")]
    with open(os.path.join(P, dst), "w") as f:
        f.write(f"<!-- {NOTE} -->\n" + (f"# {title}\n\n" if title else "") + "".join(body))


def txt(src, dst, head):
    with open(os.path.join(P, dst), "w") as f:
        f.write(f"# session {tag}: {head}\n" + open(os.path.join(G, src)).read())


open(os.path.join(P, "r06_bench_b64_fp16.json"), "w").write(last_json_line(os.path.join(G, f"{tag}_bench.json")))
open(os.path.join(P, "r06_bench_b64_fp32x.json"), "w").write(last_json_line(os.path.join(G, f"{tag}x_bench.json")))
shutil.copy(os.path.join(G, f"{tag}_configs.jsonl"), os.path.join(P, "r06_bench_configs.jsonl"))
shutil.copy(os.path.join(G, f"{tag}_kernel_stats.csv"), os.path.join(P, "r06_b64_fp16_kernel_stats.csv"))
shutil.copy(os.path.join(G, f"{tag}x_kernel_stats.csv"), os.path.join(P, "r06_b64_fp32x_kernel_stats.csv"))
head = "rocprofv3 --kernel-trace --stats over bench.py (8 steps + 2 warm-up), tools/prof_summary.py"
txt(f"{tag}_prof_summary.txt", "r06_kernel_time_split.txt", head)
txt(f"{tag}x_prof_summary.txt", "r06_fp32x_kernel_time_split.txt", head)
md(f"{tag}_pmc_table.md", "r06_pmc_top_kernels.md")
md(f"{tag}x_pmc_table.md", "r06_fp32x_pmc_top_kernels.md")
md(f"{tag}_lds_conflicts_fp16.md", "r06_lds_conflicts.md", "LDS bank-conflict share per kernel, fp16 bench step (tools/lds_conflicts.sh)")
md(f"{tag}_lds_conflicts_fp32x.md", "r06_fp32x_lds_conflicts.md", "LDS bank-conflict share per kernel, fp32x bench step (tools/lds_conflicts.sh --dtype fp32x)")
md(f"{tag}_conv_layers.md", "r06_conv_layers.md", "per-layer 3x3 conv table, fp16, B = 64 (tools/bench_layers.py 64)")
md(f"{tag}_conv_layers_fp32x.md", "r06_conv_layers_fp32x.md", "per-layer 3x3 conv table, fp32x mode, B = 64 (tools/bench_layers.py 64 --fp32x)")
txt(f"{tag}_dkv_launches.txt", "r06_dkv_launches.txt", "per-launch durations of the roofline kernel from the kernel trace of the profiled run (tools/dkv_launches.py)")
for f in glob.glob(os.path.join(G, f"{tag}_dkv_traffic_*.json")):
    d = json.load(open(f))
    key = os.path.basename(f)[len(tag) + 1:]
    json.dump(d, open(os.path.join(P, "r06_" + key), "w"), indent=1)
print("collected", tag, "->", P)
for f in (f"{tag}_bench.json", f"{tag}x_bench.json"):
    d = json.loads(last_json_line(os.path.join(G, f)))
    pg = d.get("parity_grade_path") or {}
    print(f, d["value"], d["ms_per_step"], "clock", d["clock"]["clock_mhz"], "roofline", d["roofline"]["frac"],
          {k: v["frac"] for k, v in d["roofline"]["kernels"].items()}, "parity-grade", pg.get("value"))
print(open(os.path.join(G, f"{tag}_pytest.log")).read().strip().splitlines()[-1])
