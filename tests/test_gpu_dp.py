"""GPU tests of the data-parallel exchange step (maskunet_amd/dp.py; replaces nn.DataParallel, ade_semantic.py:373).

The single-GPU boxes of the test pool cannot run RCCL across devices, so the stream ordering is tested two ways:
  * a stub collective with RCCL's semantics (the reduction runs on the backend's own stream, late; ``Work.wait()`` only makes
    the stream that is CURRENT at the call wait for it) -- fails if the divide / copy-back are not ordered behind the collective;
  * two gloo ranks sharing the one GPU run a REAL forward + backward on the two halves of a batch through the autograd hooks and
    must reproduce the mean of the two shard gradients computed without any wrapper.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _LateWork:
    """all_reduce stand-in: 'reduces' (x3) on a private stream after a long spin; wait() orders only the current stream."""

    streams_at_wait = []

    def __init__(self, flat, issue_stream):
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(issue_stream)
        with torch.cuda.stream(self.stream):
            torch.cuda._sleep(200_000_000)          # ~0.1 s: anything not ordered behind the collective runs first
            flat.mul_(3.0)
        flat.record_stream(self.stream)

    def wait(self):
        cur = torch.cuda.current_stream()
        _LateWork.streams_at_wait.append(cur)
        cur.wait_stream(self.stream)
        return True


def test_wait_is_issued_on_the_comm_stream(monkeypatch):
    import torch.distributed as dist
    import maskunet_amd
    from maskunet_amd.dp import DataParallel
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "broadcast", lambda *a, **k: None)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "stub")       # not "nccl": the sum is divided in _write_back
    monkeypatch.setattr(dist, "all_reduce", lambda flat, op=None, group=None, async_op=False: _LateWork(flat, torch.cuda.current_stream()))
    _LateWork.streams_at_wait = []
    torch.manual_seed(0)
    model = maskunet_amd.DownSample(32, 64).cuda()
    ddp = DataParallel(model, bucket_mb=0.05)
    assert ddp.world == 2 and len(ddp.buckets) > 1
    x = torch.randn(4, 32, 16, 16, device="cuda")
    for step in range(2):                       # step 0 learns the dead parameters, step 1 launches every bucket from the hooks
        model.zero_grad(set_to_none=True)
        model(x).square().mean().backward()     # reference gradients without the wrapper
        ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        model.zero_grad(set_to_none=True)
        ddp(x).square().mean().backward()
        if step == 1:
            assert all(b.work is not None for b in ddp.buckets if any(id(p) not in ddp._dead for p in b.params)), \
                "live buckets must be launched from the autograd hooks"
        ddp.finish_gradient_sync()
        torch.cuda.synchronize()
        assert _LateWork.streams_at_wait and all(s == ddp._comm_stream for s in _LateWork.streams_at_wait), \
            "Work.wait() must be called with the comm stream current"
        for n, p in model.named_parameters():
            if n in ref:
                assert torch.allclose(p.grad, ref[n] * 1.5, rtol=1e-5, atol=1e-7), f"{n}: divide/copy-back raced the collective"
            else:
                assert p.grad is None


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist, torch.nn.functional as F
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
import maskunet_amd
from maskunet_amd.dp import DataParallel, shard_batch
from oracle import maskunet_oracle as O          # deterministic parameter / input recipe only
c_out, B = 19, 4
params = O.make_params(O.unet_state_shapes(3, c_out, False), 77)
keeps = O.make_keeps(78, B)
x, labels = O.make_inputs(79, B, c_out)
def fresh():
    m = maskunet_amd.UNet(3, c_out)
    m.load_state_dict(params)
    m.cuda().train()
    m.dropout.p = 0.0
    return m
def shard_grads(m, s):
    a, b = shard_batch(B, world, s)
    m.set_keep_masks([k[a:b] for k in keeps])
    out = m(x[a:b].cuda())
    F.cross_entropy(out, labels[a:b].cuda()).backward()
# expected: every shard from the SAME initial state (per-replica BatchNorm statistics), gradients averaged over the shards
exp = None
for s in range(world):
    m = fresh()
    shard_grads(m, s)
    g = {{n: p.grad.double() for n, p in m.named_parameters() if p.grad is not None}}
    exp = g if exp is None else {{n: exp[n] + v for n, v in g.items()}}
    if s == rank:
        own_stats = {{k: v.clone() for k, v in m.state_dict().items() if "running" in k}}
exp = {{n: v / world for n, v in exp.items()}}
model = fresh()
ddp = DataParallel(model, bucket_mb=4.0)
worst = 0.0
for step in range(2):                      # the second step runs with the learned dead-parameter set (all buckets from hooks)
    model.load_state_dict(params)
    model.zero_grad(set_to_none=True)
    a, b = shard_batch(B, world, rank)
    model.set_keep_masks([k[a:b] for k in keeps])
    out = ddp(x[a:b].cuda())
    F.cross_entropy(out, labels[a:b].cuda()).backward()
    if step == 1:
        assert all(bk.work is not None for bk in ddp.buckets if any(id(p) not in ddp._dead for p in bk.params))
    ddp.finish_gradient_sync()
    for n, p in model.named_parameters():
        if n in exp:
            e = float((p.grad.double() - exp[n]).abs().max()) / max(float(exp[n].abs().max()), 1e-12)
            worst = max(worst, e)
        else:
            assert p.grad is None, n
for k, v in model.state_dict().items():     # per-replica BatchNorm: this rank's running statistics are its own shard's
    if "running" in k:
        assert torch.allclose(v, own_stats[k], rtol=1e-5, atol=1e-6), k
assert all(k.startswith("module.") for k in ddp.state_dict())
print("RANK", rank, "worst", worst, flush=True)
# identical shard runs are bit-reproducible (no float atomics), the cross-rank mean goes through fp32 on the host: ~1e-7
assert worst <= 1e-5, worst
# the same step as ONE captured graph per replica + the eager exchange after each replay (GraphedStep over DataParallel)
model.load_state_dict(params)
model.zero_grad(set_to_none=True)
step_fn = maskunet_amd.GraphedStep(ddp, F.cross_entropy, x[a:b].cuda(), labels[a:b].cuda())
worst_g = 0.0
for it in range(2):
    loss = step_fn(x[a:b].cuda(), labels[a:b].cuda())
    assert torch.isfinite(loss).all()
    for n, p in model.named_parameters():
        if n in exp:
            e = float((p.grad.double() - exp[n]).abs().max()) / max(float(exp[n].abs().max()), 1e-12)
            worst_g = max(worst_g, e)
        else:
            assert p.grad is None, n
    model.zero_grad(set_to_none=True)         # the next call re-attaches the graph's gradient tensors
print("RANK", rank, "worst graphed", worst_g, flush=True)
assert worst_g <= 1e-5, worst_g
dist.destroy_process_group()
"""


def test_two_ranks_one_gpu_real_backward_matches_shard_mean():
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(o[-3000:] for o in outs)


_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist, torch.nn.functional as F
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # exactly bench.py's call
import maskunet_amd
from maskunet_amd.dp import DataParallel
from oracle import maskunet_oracle as O          # deterministic parameter / input recipe only
c_out, B = 19, 2
params = O.make_params(O.unet_state_shapes(3, c_out, False), 77)
keeps = O.make_keeps(78, B)
x, labels = O.make_inputs(79, B, c_out)
model = maskunet_amd.UNet(3, c_out)
model.load_state_dict(params)
model.cuda().train()
model.dropout.p = 0.0
model.set_keep_masks(keeps)
xd, ld = x.cuda(), labels.cuda()
F.cross_entropy(model(xd), ld).backward()
ref = {{n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}}
ddp = DataParallel(model, bucket_mb=4.0, force_sync=True)
assert ddp.multi and ddp.overlap and len(ddp.buckets) > 3
flattened = []
_reduce = ddp._reduce
ddp._reduce = lambda params: (flattened.append(len(params)), _reduce(params))[1]      # the torch.cat path
big = [(n, p) for n, p in model.named_parameters() if n in ref and (p.dim() == 4 or n in ("norm.weight", "norm.bias"))]
for step in range(3):                      # step 0 learns the dead parameters; later steps launch every bucket from the hooks
    model.zero_grad(set_to_none=True)
    flattened.clear()
    F.cross_entropy(ddp(xd), ld).backward()
    if step:
        assert all(bk.work is not None for bk in ddp.buckets if any(id(p) not in ddp._dead for p in bk.params))
    ddp.finish_gradient_sync()
    if step:
        # zero-copy exchange (VERDICT r4 #6a): the conv / Linear weight-gradient kernels and the final LayerNorm's backward wrote into the
        # bucket slices themselves, every live p.grad IS its slice of the bucket that was all-reduced in place, and no bucket was flattened
        assert not flattened, flattened
        for n, p in model.named_parameters():
            if n in ref:
                v = ddp.gradient_slice(p)
                assert v is not None and p.grad.data_ptr() == v.data_ptr(), n
        n_big = sum(p.numel() for _, p in big)
        assert n_big > 0.97 * sum(p.numel() for n, p in model.named_parameters() if n in ref)
    for n, p in model.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), n        # sum over one rank / 1: bit-exact through the fp32 buckets
        else:
            assert p.grad is None, n
# ... and the large gradients are in their slices right after the backward, BEFORE the exchange touches anything (the kernels wrote there)
model.zero_grad(set_to_none=True)
with ddp.no_sync():
    F.cross_entropy(ddp(xd), ld).backward()
for n, p in big:
    if "query" in n or "key" in n or "value" in n:
        continue                                      # (the three Linear weights of a block come out of ONE [3C, C] kernel output: copied in)
    assert p.grad.data_ptr() == ddp.gradient_slice(p).data_ptr(), n
    assert torch.equal(p.grad, ref[n]), n
ddp.finish_gradient_sync()
model.zero_grad(set_to_none=True)
# the same through a captured step: TWO graph segments split inside the one backward pass at the encoder | bottleneck boundary, the
# buckets the first segment completes are all-reduced on the comm stream beside the second segment's replay (VERDICT r4 #6b)
step_fn = maskunet_amd.GraphedStep(ddp, F.cross_entropy, xd, ld)
assert step_fn.graph2 is not None
seg1 = {{n for n, p in model.named_parameters() if id(p) in step_fn._seg1}}
assert "norm.weight" in seg1 and any(n.startswith("bottom") for n in seg1) and any(n.startswith("upsample1") for n in seg1), sorted(seg1)[:8]
assert not any(n.startswith(("initial_conv", "downsample", "self_attention1", "self_attention2", "self_attention3")) for n in seg1)
early = []
_lcb = ddp.launch_complete_buckets
ddp.launch_complete_buckets = lambda done: (early.append(_lcb(done)), early[-1])[1]
for it in range(2):
    model.zero_grad(set_to_none=True)
    step_fn(xd, ld)
    for n, p in model.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), n
            assert p.grad.data_ptr() == ddp.gradient_slice(p).data_ptr(), n
assert early and all(k >= 2 for k in early), early          # most buckets start between the segments, before the encoder's backward replays
# the other collectives bench.py issues
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
ddp.broadcast_buffers()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL single-rank exchange ok", flush=True)
"""


def test_rccl_backend_single_rank_exchange():
    """The nccl (= RCCL) backend itself, in a group of one rank (all the single-GPU pool offers): init with device_id as bench.py does,
    async all-reduces issued from the autograd hooks on the comm stream, Work.wait() there, write-back, barrier, broadcast."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_WORKER.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "exchange ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_starts_under_torchrun_world2_gloo():
    """The driver's N>1 invocation (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with two gloo
    ranks on the one GPU: must initialise, step through DataParallel and print ONE JSON line with n_gpus = 2."""
    import json
    env = dict(os.environ, MU_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0
    # N > 1 times the reference's multi-GPU semantics: nn.DataParallel discards its replicas, so every step draws fresh key masks
    assert rec["config"]["mask_mode"] == "resample"


@pytest.mark.parametrize("graph", [False, True])
def test_bench_multi_rank_path_over_rccl_one_rank(graph):
    """bench.py's N > 1 code path (nccl init with device_id, DataParallel hooks / buckets on the comm stream, barrier, max over ranks;
    with --graph: replay + exchange) on the RCCL backend in a group of one rank, launched exactly as the driver launches N > 1."""
    import json
    env = dict(os.environ, MU_BENCH_FORCE_DP="1")
    env.pop("MU_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--batch", "8", "--no-cpu-baseline"] + (["--graph"] if graph else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["roofline"]["frac"] > 0
    assert rec["config"]["mask_mode"] == "resample"      # round 6: a captured step redraws its masks per replay too (the reference's N > 1 semantics)


def test_bench_bare_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (the shape of command a scaling driver may issue): bench.py starts the two ranks
    itself as a child torch.distributed.run (before touching the GPU) and relays rank 0's JSON line and the exit code.  Two gloo ranks
    share the one GPU here."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MU_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0
    assert rec["config"]["mask_mode"] == "resample" and rec["config"]["parallelism"] == "dp2"


def test_bench_refuses_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_bench_line_is_self_describing():
    """The N = 1 line carries what a reader of BENCH alone needs: which parity gate the timed dtype passes (the reference-generated
    B = 2 golden, run outside the timed region), the clock the chip holds under a dense MFMA load, and the roofline fraction at it."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"],
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    g = rec["parity_gate"]
    assert g["passed"] is True and g["dtype"] == "f16" and g["gate"]["out"] == 3e-2 and 0 < g["observed"]["out"] <= 3e-2
    c = rec["clock"]
    assert 800 < c["clock_mhz"] <= 2500 and c["probe_tflops"] > 200
    assert rec["roofline"]["frac_at_measured_clock"] >= rec["roofline"]["frac"] > 0
    pg = rec["parity_grade_path"]                      # the same workload in the fp32x mode, behind the timed region
    assert pg["value"] > 0 and pg["finite"] and pg["parity_gate"]["passed"] is True and pg["parity_gate"]["observed"]["out"] <= 1e-3
    # ... as a first-class measurement: >= 10 timed steps after 2 warm-up steps and a roofline block of its own (VERDICT r4 #5a)
    assert pg["steps"] >= 10 and pg["warmup"] >= 2 and 0 < pg["roofline"]["frac"] < 1 and pg["roofline"]["launches_timed"] == pg["steps"]
    # every sweep of the dominant attention block with its own executed-FLOP fraction (VERDICT r4 #5d)
    ks = rec["roofline"]["kernels"]
    assert set(ks) == {"fwd", "dq", "dkv"} and all(0 < k["frac"] < 1 and k["launches_timed"] == 2 for k in ks.values())
    assert "power_w" not in rec["clock"]["smi"] and "power_rails_w" in rec["clock"]["smi"]


def test_bench_gpus2_over_rccl_on_a_one_gpu_box_refuses_with_a_message():
    """VERDICT r4 #6c: `python bench.py --gpus 2` bare, RCCL backend, on a box with ONE GPU -- a clean non-zero exit with a message that
    names the cause, from the parent (nothing spawned, no rank dying in hipSetDevice / ncclCommInitRank)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MU_DIST_BACKEND")}
    env["MU_DIST_BACKEND"] = "nccl"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in (r.stdout + r.stderr) and "Traceback" not in r.stderr, r.stderr[-2000:]
    # the same refusal inside a launcher-started rank (WORLD_SIZE = 2 from torchrun, one device)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 2 visible GPUs" in (r.stdout + r.stderr)



_RCCL_FP32X_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist, torch.nn.functional as F
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import maskunet_amd
from maskunet_amd.dp import DataParallel
from oracle import maskunet_oracle as O          # deterministic parameter / input recipe only
maskunet_amd.set_float32_matmul_precision("high")          # fp32x: the 3x3 weight gradients come out of mu_conv_wgrad_h (round 6)
c_out, B = 19, 2
params = O.make_params(O.unet_state_shapes(3, c_out, False), 91)
keeps = O.make_keeps(92, B)
x, labels = O.make_inputs(93, B, c_out)
model = maskunet_amd.UNet(3, c_out)
model.load_state_dict(params)
model.cuda().train()
model.dropout.p = 0.0
model.set_keep_masks(keeps)
xd, ld = x.cuda(), labels.cuda()
F.cross_entropy(model(xd), ld).backward()
ref = {{n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}}
ddp = DataParallel(model, bucket_mb=4.0, force_sync=True)
for step in range(3):
    model.zero_grad(set_to_none=True)
    F.cross_entropy(ddp(xd), ld).backward()
    ddp.finish_gradient_sync()
    for n, p in model.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), (step, n)          # one rank: the mean is the gradient itself, bit for bit
            if step:
                assert p.grad.data_ptr() == ddp.gradient_slice(p).data_ptr(), n
# the two-term weight-gradient kernels wrote straight into their bucket slices (no copy in between)
model.zero_grad(set_to_none=True)
with ddp.no_sync():
    F.cross_entropy(ddp(xd), ld).backward()
n3 = 0
for n, p in model.named_parameters():
    if n in ref and p.dim() == 4 and p.shape[-1] == 3 and p.shape[1] > 3:
        assert p.grad.data_ptr() == ddp.gradient_slice(p).data_ptr(), n
        n3 += 1
assert n3 >= 30, n3
ddp.finish_gradient_sync()
dist.destroy_process_group()
print("fp32x exchange ok", flush=True)
"""


def test_rccl_single_rank_exchange_in_the_fp32x_mode():
    """The zero-copy exchange with the round-6 fp32x backward: mu_conv_wgrad_h (two-term weight gradient, pair-reduced slabs) writes the 3x3
    weight gradients into DataParallel's bucket slices like the kernels of the other modes; gradients bit-equal to the plain model's."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_FP32X_WORKER.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fp32x exchange ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
