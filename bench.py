#!/usr/bin/env python3
"""Benchmark of the MaskAttn-UNet forward+backward hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): 128x128 images/sec, forward + backward, whole job (all N GPUs).
Workload at N=1: configs[1] = "ADE20K semantic 128x128, batch=64, 1xMI355X fp16": UNet(3,150), B=64 per GPU,
fp16 storage / fp32 accumulate, synthetic images/labels/key-masks, random-init weights, train mode (batch-stat
BatchNorm, dropout 0.3), loss = mean pixel cross-entropy on the module's NCHW fp32 output (maskunet_amd.CrossEntropyLoss,
the reference's criterion as HIP kernels; --torch-loss = torch's F.cross_entropy) with a static loss scale
for the fp16 backward.  N>1: one process per GPU,
the same per-GPU batch (weak scaling), one bucketed RCCL all-reduce of the gradients per step overlapped with the
backward (maskunet_amd/dp.py).

One JSON line on rank 0.  `roofline` is for the dominant kernel group (the N=16384 attention block), its duration
measured live with HIP events on the launch stream during the timed steps.  `cpu_baseline` is the CPU oracle
(oracle/maskunet_oracle.py == restatement of the reference's PyTorch CPU path) timed on this host's cores on a
bounded sample (B=1) -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def synth(B, c_out, hw, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((B, 3, hw, hw), generator=g)
    y = torch.randint(0, c_out, (B, hw, hw), generator=g)
    ns = [(hw // 2) ** 2, (hw // 4) ** 2, (hw // 8) ** 2, (hw // 4) ** 2, (hw // 2) ** 2, hw ** 2]
    keeps = [torch.randint(0, 2, (B, n), generator=g, dtype=torch.uint8) for n in ns]
    return x.to(device), y.to(device), [k.to(device) for k in keeps]


def cpu_baseline(c_out, hw, seconds_budget=20.0):
    """Oracle (kind 'port') fwd+bwd, fp32, train mode, on all host cores; bounded sample B=1."""
    from oracle import maskunet_oracle as O
    # torch's intra-op scaling collapses with hundreds of threads on this path (256 threads on the GPU box's
    # host: 232 s/iteration); 32 threads is the fastest setting found, `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B = 1
    shapes = O.unet_state_shapes(3, c_out, False, hw=hw)
    p = O.make_params(shapes, 1)
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    keeps = O.make_keeps(2, B, hw)
    x, labels = O.make_inputs(3, B, c_out, hw)
    times = []
    t_all = time.time()
    for it in range(3):
        t0 = time.time()
        out = O.unet_forward(p, x, keeps, training=True)
        loss = O.pixel_cross_entropy(out, labels)
        loss.backward()
        for v in p.values():
            v.grad = None
        times.append(time.time() - t0)
        if it >= 1 and time.time() - t_all > seconds_budget:
            break
    steady = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
    return {"value": round(B / steady, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd, fp32, train mode, B={B}, c_out={c_out}, {hw}x{hw}, median of {max(len(times) - 1, 1)} after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--c-out", type=int, default=150)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--three-head", action="store_true")
    ap.add_argument("--fused-loss", action="store_true",
                    help="SURVEY 8-f1 path: fused NHWC cross-entropy on the internal logits instead of module output + torch CE")
    ap.add_argument("--torch-loss", action="store_true", help="torch's F.cross_entropy on the module output instead of maskunet_amd.CrossEntropyLoss")
    ap.add_argument("--graph", action="store_true", help="replay the step (forward + criterion + backward) as one HIP graph "
                    "(maskunet_amd.GraphedStep; single GPU)")
    ap.add_argument("--optimizer", action="store_true", help="also run the fused AdamW step (8-f2) inside the timed step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MU_DIST_BACKEND", "nccl")      # "gloo" lets two ranks share one GPU in a debug run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            local = local % max(ndev, 1)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import maskunet_amd
    from maskunet_amd import _lib
    dtype = torch.float16 if args.dtype == "fp16" else torch.float32
    torch.manual_seed(1234)
    model = (maskunet_amd.InstanceUNet(3, args.c_out, 16, hw=args.hw) if args.three_head
             else maskunet_amd.UNet(3, args.c_out, hw=args.hw)).to(dev)
    model.set_compute_dtype(dtype).train()
    x, labels, keeps = synth(args.batch, args.c_out, args.hw, 42 + rank, dev)
    model.set_keep_masks(keeps)
    net = maskunet_amd.DataParallel(model) if world > 1 else model
    scale = args.loss_scale if dtype == torch.float16 else 1.0

    inst_loss = inst_labels = None
    if args.three_head:              # synthetic instance ids: 16x16 blocks of ids 0..20 (0 = background), device-side loss (row f1b)
        g = torch.Generator().manual_seed(7 + rank)
        blocks = torch.randint(0, 21, (args.batch, args.hw // 16, args.hw // 16), generator=g)
        inst_labels = blocks.repeat_interleave(16, 1).repeat_interleave(16, 2).to(dev)
        inst_loss = maskunet_amd.InstanceContrastiveLoss(margin=1.0, ignore_index=255)
    criterion = maskunet_amd.CrossEntropyLoss()      # the reference's nn.CrossEntropyLoss() on the module output (ade_semantic.py:377,399)
    opt = maskunet_amd.FusedAdamW(model.parameters(), lr=5e-5, weight_decay=1e-1) if args.optimizer else None

    graphed = None
    if args.graph:
        if world > 1 or args.fused_loss or args.three_head or args.torch_loss:
            # N > 1 is not offered: a replay runs no autograd hooks (the bucket all-reduces would follow it instead of overlapping the
            # backward), and the only multi-rank rig of this round -- two gloo ranks sharing one GPU -- ran it 10x slower than eager
            raise SystemExit("--graph: single GPU, maskunet_amd.CrossEntropyLoss, 1-head model only")
        graphed = maskunet_amd.GraphedStep(model, criterion, x, labels, loss_scale=scale)

    def step():
        nonlocal graphed
        if graphed is not None:
            loss = graphed(x, labels)
            if opt is not None:
                opt.step(grad_scale=scale)
            return loss
        if args.fused_loss and not args.three_head:
            if world > 1:
                net._arm()
            loss = maskunet_amd.pixel_cross_entropy_nhwc(model.logits_nhwc(x), labels, args.c_out, grad_scale=scale)
            loss.backward()
        else:
            out = net(x)
            sem = out[0] if args.three_head else out
            loss = F.cross_entropy(sem, labels) if args.torch_loss else criterion(sem, labels)
            if args.three_head:          # city_instance.py:372-377: seg_loss + LAMBDA_IE * InstanceContrastiveLoss(embeddings, inst_labels)
                loss = loss + 0.1 * inst_loss(out[2], inst_labels)
            (loss * scale).backward()
        if world > 1:
            net.finish_gradient_sync()
        if opt is not None:
            opt.step(grad_scale=scale)
        model.zero_grad(set_to_none=True)
        return loss

    for _ in range(args.warmup):
        step()
    # HIP-event probe on the dominant kernel: the dK/dV sweep (attn_bwd_dkv3_kernel) of self_attention6, N = hw*hw.
    # mu_attn_bwd_phases(..., B, N, C, nkmax, ws, ws_bytes, dtype, phases, stream): args[16] = N, args[22] = phases
    N6 = args.hw * args.hw
    _lib.PROBE = {"pred": lambda name, a: name == "mu_attn_bwd_phases" and a[16] == N6 and a[22] == 4, "events": []}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    probe = _lib.PROBE
    if graphed is not None:                     # a replayed graph makes no Python calls to probe: time the kernel in two eager steps
        graphed = None                          # after the timed region (on every rank: eager steps all-reduce)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    _lib.PROBE = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.isfinite(loss.detach()).all():
        raise RuntimeError("non-finite loss in the timed region")

    if rank == 0:
        imgs = args.batch * world * args.steps
        C6 = 64
        # algorithmic FLOPs of one launch: the four N x N x C products of the dK/dV sweep (S, dP, dV, dK) over the FULL
        # key set, 4 * 2*N*N*C per image; the kernel executes ~half (masked keys are skipped)
        flops = 8.0 * N6 * N6 * C6 * args.batch
        durs = [e0.elapsed_time(e1) * 1e-3 for _, e0, e1 in probe["events"]]
        tk = sum(durs) / max(len(durs), 1)
        achieved = flops / max(tk, 1e-12) / 1e12
        kept = [float(k.float().mean().item()) for k in keeps if k.shape[1] == N6]
        kept = kept[-1] if kept else 1.0     # fraction of keys the kernel really multiplies (the rest are skipped, not computed)
        peak = PEAK_MFMA_TFLOPS[args.dtype]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_dkv_traffic.json")      # PMC FETCH_SIZE/WRITE_SIZE pass (see file)
        if os.path.exists(tpath) and args.batch == 64 and args.dtype == "fp16":
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        rec = {
            "metric": "128x128 images/sec (fwd+bwd)", "value": round(imgs / elapsed, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"ADE20K-semantic shape {args.hw}x{args.hw}, c_out={args.c_out}, batch={args.batch}/GPU, "
                                   f"{'3-head' if args.three_head else '1-head'} MaskAttn-UNet fwd+bwd, train mode",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "loss_scale": scale,
                       "loss": "fused NHWC CE kernel" if args.fused_loss else ("torch CE on module output" if args.torch_loss else
                                                                                "maskunet_amd.CrossEntropyLoss on module output"), "optimizer_in_step": bool(opt), "hip_graph": bool(args.graph)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "kernel": f"attn_bwd_dkv3_kernel (self_attention6 dK/dV sweep, N={N6}, C=64)",
                         "ms_per_launch": round(tk * 1e3, 3), "launches_timed": len(durs),
                         "kept_keys": round(kept, 4), "executed_achieved": round(achieved * kept, 2),
                         "executed_frac": round(achieved * kept / peak, 4),
                         "note": "achieved = algorithmic FLOPs (8*N*N*C per image over the FULL key set, SURVEY 8-d4) / time; the kernel "
                                 "skips masked keys, executed_* count only the products it really performs"},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.c_out, args.hw)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
