#!/usr/bin/env python3
"""Whole-step A/B of environment switches on ONE box (debug aid): python tools/ab_env.py NAME[:VAR=V,VAR2=V2] ... [-- bench.py args]
Every arm runs bench.py in its own process with its variables set (MU_LIB_PATH=... selects a library variant too); rounds interleaved,
median ms/step per arm (cdna guide rule 24: never compare timings taken on different boxes)."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); args, extra = args[:i], args[i + 1:]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
arms = []
for a in args:
    name, _, kv = a.partition(":")
    env = dict(os.environ)
    for item in filter(None, kv.split(",")):
        k, _, v = item.partition("=")
        env[k] = v
    arms.append((name, env))
res = {n: [] for n, _ in arms}
for rnd in range(rounds):
    for n, env in arms:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "3"] + extra,
                             capture_output=True, text=True, timeout=900, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "failed:", out.stderr[-600:]); continue
        res[n].append(json.loads(line[-1])["ms_per_step"])
for n, v in res.items():
    if v:
        print(f"{n}: ms/step median {statistics.median(v):.3f}  all {v}")
