"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_summary.py <csv> [name-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ncall = collections.defaultdict(set)
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        ncall[r["Kernel_Name"][:60]].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(ncall[k])
    print(f"{k}  ({n} dispatches)")
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    for c, x in sorted(v.items()):
        extra = f"  {100 * x / wc:5.1f}% of WAVE_CYCLES" if wc and c != "SQ_WAVE_CYCLES" and c.startswith("SQ_") else ""
        print(f"    {c:32s} {x / n:16.0f} /dispatch{extra}")
