#!/usr/bin/env python3
"""In-process timing of the data-parallel code path in a group of ONE rank (debug aid): plain step vs DataParallel with / without overlap.
python tools/ab_dp.py   (single GPU; RCCL backend)"""
import os, sys, time, statistics
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import maskunet_amd
import bench

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = maskunet_amd.UNet(3, 150).to(dev)
model.set_compute_dtype(torch.float16).train()
x, labels, keeps = bench.synth(64, 150, 128, 42, dev)
model.set_keep_masks(keeps)
crit = maskunet_amd.CrossEntropyLoss()

def make(kind):
    if kind == "plain":
        net = model
    else:
        net = maskunet_amd.DataParallel(model, force_sync=True, overlap=(kind != "dp_nooverlap"), bucket_mb=(8.0 if kind == "dp_8mb" else 32.0))
    def step():
        out = net(x)
        (crit(out, labels) * 1024.0).backward()
        if net is not model:
            net.finish_gradient_sync()
        model.zero_grad(set_to_none=True)
    return net, step

kinds = ["plain", "dp", "dp_nooverlap", "dp_8mb"]
steps = {}
nets = {}
for k in kinds:
    nets[k], steps[k] = make(k)
    for _ in range(3): steps[k]()
    # hooks of a DataParallel stay registered on the parameters: drop them before the next variant
    if nets[k] is not model:
        for h in nets[k]._hooks: h.remove()
res = {k: [] for k in kinds}
for rnd in range(4):
    for k in kinds:
        net, _ = make(k)
        st = steps[k] = make(k)[1] if False else None
        net, st = make(k)
        for _ in range(2): st()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): st()
        torch.cuda.synchronize(); res[k].append((time.perf_counter() - t0) / 8 * 1e3)
        if net is not model:
            for h in net._hooks: h.remove()
for k, v in res.items():
    print(f"{k}: ms/step median {statistics.median(v):.3f}  all {[round(t, 3) for t in v]}")
dist.destroy_process_group()
