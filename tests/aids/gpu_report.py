#!/usr/bin/env python3
"""Run every GPU parity check without stopping at the first failure and print a table (debug aid)."""
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _gpu_checks as G  # noqa: E402


def run(title, fn, *a, **k):
    t = time.time()
    try:
        res = fn(*a, **k)
        torch.cuda.synchronize()
    except Exception:
        print(f"[EXC ] {title}\n{traceback.format_exc()}")
        return
    bad = [r for r in res if not (r[1] <= r[2])]
    print(f"[{'FAIL' if bad else 'ok  '}] {title}: {len(res)} checks, worst {max(r[1] for r in res):.3e}, {time.time() - t:.1f}s")
    for n, e, tol in (res if os.environ.get('VERBOSE') else bad):
        print(f"        {n}: err {e:.3e} tol {tol:.1e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["kernels", "modules", "unet"]
    for dt in (torch.float32, torch.float16):
        tag = str(dt).split(".")[-1]
        if "kernels" in which:
            run(f"transpose {tag}", G.check_transpose, dt)
            run(f"layout {tag}", G.check_layout_roundtrip, dt)
            run(f"prep_weight {tag}", G.check_prep_weight, dt)
            run(f"conv_stats {tag}", G.check_conv_stats, dt)
            run(f"conv {tag}", G.check_conv, dt)
            run(f"bn_act {tag}", G.check_bn_act, dt)
            run(f"pool/up {tag}", G.check_pool_up, dt)
            run(f"ln_sample {tag}", G.check_ln_sample, dt)
            run(f"dropout {tag}", G.check_dropout, dt)
            run(f"attention {tag}", G.check_attention, dt)
        if "modules" in which:
            import glob
            for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "golden", "*.npz"))):
                n = os.path.basename(p)[:-4]
                if n.startswith(("unet", "up_32_16")):
                    continue
                run(f"golden {n} {tag}", G.check_golden_module, n, dt)
        if "unet" in which:
            for n in ("unet1_c150_b2_eval", "unet1_c150_b2_train", "unet3_c19_b2_train"):
                run(f"golden {n} {tag}", G.check_unet_golden, n, dt)
            run(f"unet vs oracle {tag}", G.check_unet_vs_oracle, dt)
    if "kernels" in which:
        run("mask semantics", G.check_attention_mask_semantics)
