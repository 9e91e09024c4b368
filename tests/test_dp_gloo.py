"""CPU, world_size 2, gloo: the data-parallel exchange step (maskunet_amd/dp.py).  The all-reduce logic is
device-agnostic, so it is exercised here with real parameters of the model and synthetic gradients (the HIP
forward itself needs a GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maskunet_amd.dp import DataParallel, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import maskunet_amd
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        model = maskunet_amd.DownSample(32, 64)
        ddp = DataParallel(model, bucket_mb=0.05, overlap=overlap)
        # replicas must be identical after construction (rank 0's weights)
        ref = [torch.zeros_like(p) for p in model.parameters()]
        for r, p in zip(ref, model.parameters()):
            r.copy_(p.data)
            dist.broadcast(r, 0)
            assert torch.equal(r, p.data), "parameters not broadcast from rank 0"
        assert len(ddp.buckets) > 1
        emb = [p for n, p in model.named_parameters() if "emb_layer" in n]

        def one_step(revive=False):
            """emulate autograd: accumulate a rank-dependent gradient per live parameter, fire the post-accumulate hook"""
            ddp._arm()
            expected = {}
            for name, p in model.named_parameters():
                p.grad = None
                if "emb_layer" in name and not revive:         # dead weights: never a gradient
                    continue
                g0 = torch.full_like(p, 1.0) * (len(name) % 7 + 1)
                expected[name] = g0 * 1.5                      # mean of g0*(rank+1) over ranks 0,1
                p.grad = g0 * (rank + 1)
                if overlap:
                    ddp._on_grad(p)
            launched = sum(b.work is not None for b in ddp.buckets)
            ddp.finish_gradient_sync()
            for name, p in model.named_parameters():
                if name in expected:
                    assert torch.allclose(p.grad, expected[name]), name
                else:
                    assert p.grad is None
            return launched

        first = one_step()
        assert ddp._dead == {id(p) for p in emb}, "parameters without a gradient must be learned on the first step"
        second = one_step()
        # zero-copy exchange (round 5): from the second step every live parameter's gradient IS its slice of the bucket's persistent
        # flat buffer (the arena), which was all-reduced in place -- no flattening copy, no write-back, a stable p.grad identity
        for name, p in model.named_parameters():
            if "emb_layer" in name:
                assert ddp.gradient_slice(p) is None
                continue
            v = ddp.gradient_slice(p)
            bk = ddp._bucket_of[id(p)]
            assert v is not None and p.grad.data_ptr() == v.data_ptr(), name
            lo, hi = bk.arena.data_ptr(), bk.arena.data_ptr() + bk.arena.numel() * 4
            assert lo <= p.grad.data_ptr() and p.grad.data_ptr() + p.grad.numel() * 4 <= hi, name
        ptrs = {n: p.grad.data_ptr() for n, p in model.named_parameters() if p.grad is not None}
        third = one_step()
        assert {n: p.grad.data_ptr() for n, p in model.named_parameters() if p.grad is not None} == ptrs and third == second
        if overlap:
            # from the second step on no bucket waits for a dead parameter: every bucket with a live gradient is launched
            # from the hooks, before finish_gradient_sync()
            live_buckets = sum(any(id(p) not in ddp._dead for p in b.params) for b in ddp.buckets)
            assert second == live_buckets and second >= first, (first, second, live_buckets)
        one_step(revive=True)                                   # a parameter believed dead gets a gradient: still averaged
        assert not ddp._dead
        one_step()
        with ddp.no_sync():                                     # accumulation step: nothing is reduced
            ddp._arm()
            for p in model.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
                if overlap:
                    ddp._on_grad(p)
            assert all(b.work is None for b in ddp.buckets)
            ddp.finish_gradient_sync()
            assert all(torch.equal(p.grad, torch.full_like(p, float(rank + 1))) for p in model.parameters())
        # state_dict keys carry the nn.DataParallel "module." prefix
        assert all(k.startswith("module.") for k in ddp.state_dict().keys())
        # the arena's registration with ops goes away with the wrapper (no leak when wrappers are re-created)
        from maskunet_amd import ops
        assert ops.GRAD_ARENA and sum(id(p) in ops.GRAD_ARENA for p in model.parameters()) >= 10
        ddp.close()
        assert not any(id(p) in ops.GRAD_ARENA for p in model.parameters())
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _run(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_gradient_allreduce_world2_overlap():
    _run(True)


def test_gradient_allreduce_world2_deferred():
    _run(False)


def test_shard_batch_partitions_like_scatter():
    for gb, w in [(512, 8), (256, 8), (10, 4), (7, 2)]:
        spans = [shard_batch(gb, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _single_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import maskunet_amd
        torch.manual_seed(3)
        model = maskunet_amd.DownSample(32, 64)
        plain = DataParallel(model, bucket_mb=0.05)
        assert not plain.multi and not plain.overlap                  # a group of one rank: no exchange by default
        ddp = DataParallel(model, bucket_mb=0.05, force_sync=True)
        assert ddp.multi and ddp.overlap and len(ddp.buckets) > 1
        for step in range(2):
            ddp._arm()
            exp = {}
            for name, p in model.named_parameters():
                p.grad = None
                if "emb_layer" in name:
                    continue
                p.grad = torch.full_like(p, float(len(name) % 5 + 1))
                exp[name] = p.grad.clone()
                ddp._on_grad(p)
            if step:
                assert all(b.work is not None for b in ddp.buckets if any(id(p) not in ddp._dead for p in b.params))
            ddp.finish_gradient_sync()
            for name, p in model.named_parameters():
                if name in exp:
                    assert torch.equal(p.grad, exp[name]), name       # mean over one rank
                else:
                    assert p.grad is None
        q.put("ok")
    except Exception as e:                                             # noqa: BLE001
        q.put(repr(e))
    finally:
        dist.destroy_process_group()


def test_force_sync_runs_the_exchange_in_a_group_of_one_rank():
    """DataParallel(force_sync=True): hooks, buckets, collectives and write-back with world_size 1 (what the single-GPU boxes use to
    exercise the RCCL path, tests/test_gpu_dp.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert q.get(timeout=5) == "ok"


def _advice_r5_worker(rank, world, port, q):
    """(1) a half / bf16 model through the arena-enabled wrapper, (2) copy_back=True with caller-held gradient tensors on the
    zero-copy buckets, (3) a parameter feeding two autograd nodes of one backward pass must not get its slice twice."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn as nn
        from maskunet_amd import ops
        # (1) bf16 parameters: their gradients cannot be fp32 arena views -- such buckets keep the flatten / write-back path
        torch.manual_seed(7)
        mixed = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4).to(torch.bfloat16))
        ddp = DataParallel(mixed, bucket_mb=1e-4)                      # one parameter per bucket
        for step in range(4):
            mixed.zero_grad(set_to_none=True)
            x = torch.full((3, 8), float(rank + 1))
            y = mixed[1](mixed[0](x).to(torch.bfloat16)).float().sum()
            ddp._arm()
            y.backward()
            ddp.finish_gradient_sync()
            for p in mixed.parameters():
                assert p.grad is not None and p.grad.dtype == p.dtype
                g = p.grad.float().clone()
                dist.all_reduce(g)
                assert torch.allclose(g / world, p.grad.float(), rtol=1e-2), "ranks disagree on the averaged gradient"
        assert all(ddp.gradient_slice(p) is None for p in mixed[1].parameters())          # bf16: no slice
        assert all(ddp.gradient_slice(p) is not None for p in mixed[0].parameters())      # fp32: zero-copy
        ddp.close()
        # (2) copy_back=True: the averaged values land in the tensors the caller holds, p.grad stays bound to them
        lin = nn.Linear(6, 5)
        ddp = DataParallel(lin, bucket_mb=1e-4)
        held = {n: torch.zeros_like(p) for n, p in lin.named_parameters()}
        for step in range(3):
            ddp._arm()
            for n, p in lin.named_parameters():
                held[n].fill_(float(rank + 1) * (step + 1))
                p.grad = held[n]
                ddp._on_grad(p)
            ddp.finish_gradient_sync(copy_back=True)
            for n, p in lin.named_parameters():
                assert p.grad is held[n], (step, n)
                assert torch.equal(held[n], torch.full_like(held[n], 1.5 * (step + 1))), (step, n, held[n].flatten()[:2])
        ddp.close()
        # (3) one slice per parameter per step from ops.grad_out
        lin = nn.Linear(4, 4)
        ddp = DataParallel(lin, bucket_mb=1.0)
        ddp._arm()
        for p in lin.parameters():
            p.grad = torch.ones_like(p)
            ddp._on_grad(p)
        ddp.finish_gradient_sync()                                      # the arena exists from the next _arm on
        lin.zero_grad(set_to_none=True)
        ddp._arm()
        w = lin.weight
        a = ops.grad_out(w, tuple(w.shape), w.device)
        b = ops.grad_out(w, tuple(w.shape), w.device)                  # the second autograd node of the same backward pass
        assert a.data_ptr() == ddp.gradient_slice(w).data_ptr() and b.data_ptr() != a.data_ptr()
        ddp._arm()                                                      # the next step: the slice is available again
        assert ops.grad_out(w, tuple(w.shape), w.device).data_ptr() == a.data_ptr()
        ddp.close()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_mixed_dtype_buckets_copy_back_and_one_slice_per_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_advice_r5_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
