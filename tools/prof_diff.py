#!/usr/bin/env python3
"""Debug aid: per-kernel time per step of two rocprofv3 kernel_stats.csv files side by side: python tools/prof_diff.py A.csv B.csv STEPS"""
import csv, sys
def load(p, steps):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps)
    return d
a, b, steps = load(sys.argv[1], float(sys.argv[3])), load(sys.argv[2], float(sys.argv[3])), float(sys.argv[3])
names = sorted(set(a) | set(b), key=lambda n: -abs(b.get(n, (0, 0))[0] - a.get(n, (0, 0))[0]))
print(f"total A {sum(v[0] for v in a.values()):.3f} ms/step, B {sum(v[0] for v in b.values()):.3f} ms/step")
for n in names[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    ta, ca = a.get(n, (0, 0)); tb, cb = b.get(n, (0, 0))
    print(f"{tb - ta:+8.3f} ms  A {ta:7.3f} ({ca:5.1f} calls)  B {tb:7.3f} ({cb:5.1f} calls)  {n[:110]}")
