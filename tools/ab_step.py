#!/usr/bin/env python3
"""Whole-step A/B of library variants on ONE box (debug aid): python tools/ab_step.py A B [C ...] [-- bench.py args]
Each variant (gpurun_variants/libmu_<NAME>.so, see tools/build_variant.sh) runs bench.py in its own process, rounds interleaved."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); args, extra = args[:i], args[i + 1:]
code = ("import sys, runpy; sys.path.insert(0, {root!r}); import ctypes, maskunet_amd._lib as L; L.LIB_PATH = {lib!r}; _c = ctypes.CDLL({lib!r}); "
        "L.SIGNATURES = {{k: v for k, v in L.SIGNATURES.items() if hasattr(_c, k)}}; "
        "sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '8', '--warmup', '2'] + {extra!r}; runpy.run_path({bench!r}, run_name='__main__')")
res = {n: [] for n in args}
for rnd in range(3):
    for n in args:
        lib = os.path.join(ROOT, "gpurun_variants", f"libmu_{n}.so")
        out = subprocess.run([sys.executable, "-c", code.format(root=ROOT, lib=lib, extra=extra, bench=os.path.join(ROOT, "bench.py"))],
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "failed:", out.stderr[-400:]); continue
        res[n].append(json.loads(line[-1])["ms_per_step"])
for n, v in res.items():
    print(f"{n}: ms/step median {statistics.median(v):.3f}  all {v}")
