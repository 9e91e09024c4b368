"""Data parallelism for the MaskAttn-UNet path: one process per GPU, replicated weights, batch sharded on
dim 0, ONE exchange per step = bucketed all-reduce (mean) of the parameter gradients over RCCL/xGMI.

Replaces the reference's ``torch.nn.DataParallel(model)`` (code/ade20k/ade_semantic.py:373), whose implicit
per-forward broadcast / scatter / gather / reduce-add collapses to a single gradient all-reduce:
  * BatchNorm statistics stay per replica (the reference has no SyncBN) -- faithful semantics.  Running statistics
    therefore differ between ranks; a checkpoint written by rank 0 (``state_dict()`` on rank 0, the usual recipe)
    holds RANK 0's running statistics, exactly what the reference saves (nn.DataParallel keeps replica 0's buffers,
    SURVEY 8-e1).  ``broadcast_buffers()`` copies rank 0's buffers to every rank where all ranks must evaluate
    with the same statistics;
  * parameters that never receive gradients (``emb_layer``; ``boundary_head`` under the reference loss)
    are skipped identically on every rank.  They are learned on the first step (parameters whose ``.grad`` is still
    None after the backward) and from then on are not waited for, so every bucket launches from the hooks;
  * buckets are launched from post-accumulate-grad hooks as soon as all their live gradients exist, on a side
    stream, so the all-reduce of late layers overlaps the backward of early layers.
The collective backend is whatever ``torch.distributed`` was initialised with: "nccl" (= RCCL) on GPUs,
"gloo" in the CPU tests.

Zero-copy exchange (round 5, ``arena=True``): once the dead parameters are known (after the first synchronised step) every bucket
owns ONE persistent flat fp32 buffer and every live parameter a slice of it.  The kernels that produce the large gradients write
into the slice directly (ops.GRAD_ARENA / ops.grad_out: conv and Linear weight gradients, the final LayerNorm's affine tensors --
98 % of the elements), autograd adopts that tensor as ``p.grad``, the few small ones (BatchNorm / bias vectors) are moved in by one
multi-tensor copy per bucket, and the bucket is all-reduced IN PLACE: no flattening ``torch.cat`` (a read + write of the 99.7 MB of
gradients per step), no write-back, and ``p.grad`` never changes identity afterwards.

Stream order on GPUs (the part a blocking CPU backend cannot test): gradients are produced on the current (main)
stream; the comm stream waits for the main stream, flattens the bucket and issues the all-reduce; ``Work.wait()`` is
called WITH THE COMM STREAM CURRENT, so the divide (and, with ``copy_back``, the copy into the existing ``p.grad``) are
ordered behind the collective; ``p.grad`` otherwise becomes the parameter's slice of the reduced bucket; the main stream
finally waits for the comm stream.
"""
from __future__ import annotations

import contextlib
from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn


def shard_batch(global_batch: int, world_size: int, rank: int):
    """[start, stop) of this rank's slice of a global batch (dim 0), remainder spread over the first ranks --
    the split nn.DataParallel's scatter performs."""
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class _Bucket:
    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.pending = 0
        self.flat = None
        self.work = None
        self.included: List[nn.Parameter] = []
        self.arena = None               # persistent flat fp32 buffer of the live parameters' gradients (zero-copy exchange)
        self.views = {}                 # id(p) -> its slice of `arena`, shaped like p
        self.live: List[nn.Parameter] = []
        self.in_place = False           # this step's reduce runs on `arena` itself
        self.held = []                  # (parameter, the gradient tensor it held before the in-place reduce rebound p.grad to its slice)


class DataParallel(nn.Module):
    """``model = DataParallel(model)``; use like the wrapped module; call ``finish_gradient_sync()`` (or
    ``sync_gradients()``) after ``loss.backward()`` and before ``optimizer.step()``.  ``state_dict`` keys carry the
    ``module.`` prefix exactly like nn.DataParallel checkpoints (ade_semantic.py:412; stripped on load at
    ade_panoptic.py:434)."""

    def __init__(self, module: nn.Module, bucket_mb: float = 32.0, overlap: bool = True, process_group=None,
                 force_sync: bool = False, arena: bool = True, tail_mb: float = 2.0):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_sync: run the whole exchange (broadcast, hooks, buckets, collectives) in a group of ONE rank too -- lets a
        # single-GPU box exercise the RCCL calls and their stream ordering (tests/test_gpu_dp.py)
        self.multi = self.world > 1 or (force_sync and dist.is_initialized())
        self.overlap = overlap and self.multi
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        if self.multi:
            self.broadcast_parameters()
        # buckets in reverse registration order (roughly the order gradients become ready)
        params = [p for p in module.parameters() if p.requires_grad]
        self.buckets: List[_Bucket] = []
        cur, cur_bytes, cap = [], 0, int(bucket_mb * (1 << 20))
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= cap:
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        # The LAST bucket's all-reduce cannot start before the backward has finished (it holds the first layers' gradients), so its
        # whole duration is exposed at the end of the step: peel the parameters whose gradients complete last (up to `tail_mb`) off
        # into a small bucket of their own -- the big part launches while the first layers' backward still runs, and what is left
        # exposed is a latency-bound all-reduce of a megabyte or two (UNet(3, 150): 21.6 MB -> 19.9 + 1.7 MB).
        tail_cap = int(tail_mb * (1 << 20))
        if self.buckets and tail_cap > 0:
            last = self.buckets[-1].params
            n, nb = 0, 0
            while n < len(last) - 1 and nb + last[-1 - n].numel() * 4 <= tail_cap:
                nb += last[-1 - n].numel() * 4
                n += 1
            if 0 < n < len(last):
                self.buckets[-1:] = [_Bucket(last[:-n]), _Bucket(last[-n:])]
        self._bucket_of = {id(p): b for b in self.buckets for p in b.params}
        self._comm_stream = None
        self._hooks = []
        if self.overlap:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._armed = False
        self._sync = True
        # ids of parameters that produced no gradient in the last synchronised step (None until one step has been seen)
        self._dead = None
        self._use_arena = bool(arena)
        self._arena_dead = None         # the dead set the current arena was laid out for (None = no arena)

    # ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src: int = 0):
        """Replicas start identical (nn.DataParallel re-broadcasts GPU0's parameters/buffers every forward)."""
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src, group=self.group)
        # the write above goes through `.data`, which autograd's version counter does not see: bump it so that the weight layouts
        # cached for no-grad forwards (ops._prep_cached, keyed on version + address) are rebuilt from the broadcast values
        for p in self.module.parameters():
            torch.autograd.graph.increment_version(p)
            if hasattr(p, "_mu_prep"):
                p._mu_prep.clear()

    def broadcast_buffers(self, src: int = 0):
        """Copy rank ``src``'s buffers (BatchNorm running statistics / step counters) to every rank."""
        if self.multi:
            for t in self.module.buffers():
                dist.broadcast(t.data, src, group=self.group)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context launch no all-reduce; the first synchronised
        ``finish_gradient_sync()`` afterwards reduces the accumulated gradients."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def forward(self, *args, **kwargs):
        self._arm()
        return self.module(*args, **kwargs)

    def _arm(self):
        from . import ops
        ops.GRAD_HANDED.clear()         # a new step: every slice may be handed to ONE gradient kernel again (ops.grad_out)
        dead = self._dead or ()
        if self.multi and self._use_arena and self._dead is not None and self._arena_dead != self._dead:
            self._build_arena()
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if id(p) not in dead)
            b.flat, b.work, b.included, b.in_place, b.held = None, None, [], False, []
        self._armed = self._sync

    def _build_arena(self):
        """One persistent flat fp32 gradient buffer per bucket, one slice per live parameter (bucket order), registered with ops so the
        gradient kernels write into the slices.  Rebuilt when the set of dead parameters changes."""
        from . import ops
        self._drop_arena()
        for b in self.buckets:
            b.live = [p for p in b.params if id(p) not in self._dead]
            if not b.live:
                continue
            # the slices are fp32 views that become p.grad: a parameter of another dtype cannot adopt one (autograd refuses a
            # gradient whose dtype differs from the parameter's), nor one on another device -- such a bucket keeps the flatten /
            # write-back path of _reduce / _write_back, which converts on the way in and copies back on the way out
            dev0 = b.live[0].device
            if any(p.dtype != torch.float32 or p.device != dev0 for p in b.live):
                b.live = []
                continue
            b.arena = torch.zeros(sum(p.numel() for p in b.live), dtype=torch.float32, device=b.live[0].device)
            off = 0
            for p in b.live:
                b.views[id(p)] = b.arena[off:off + p.numel()].view(p.shape)
                off += p.numel()
        if ops.GRAD_ARENA is None:
            ops.GRAD_ARENA = {}
        for b in self.buckets:
            ops.GRAD_ARENA.update(b.views)
        self._arena_dead = set(self._dead)

    def _drop_arena(self):
        from . import ops
        for b in self.buckets:
            if ops.GRAD_ARENA is not None:
                for k in b.views:
                    ops.GRAD_ARENA.pop(k, None)
            b.arena, b.views, b.live = None, {}, []
        self._arena_dead = None

    def close(self):
        """Release the arena (its slices stay registered with ops while this wrapper lives: ~4 bytes per parameter).  Called on
        garbage collection too; gradients that ARE slices keep their bucket alive on their own."""
        try:
            self._drop_arena()
        except Exception:                # interpreter shutdown: modules may be gone
            pass

    def __del__(self):
        self.close()

    def prepare_arena(self):
        """Lay the arena out NOW from the gradients that exist (parameters whose ``.grad`` is None count as dead) instead of after the
        first synchronised step -- GraphedStep calls this after its eager warm-up so that the captured kernels write into the slices."""
        if self.multi and self._use_arena:
            self._dead = {id(p) for b in self.buckets for p in b.params if p.grad is None}
            self._build_arena()

    def launch_complete_buckets(self, done_ids):
        """Start the all-reduce of every not-yet-launched bucket whose live parameters all have their gradients (ids in `done_ids`) --
        GraphedStep calls this between its two graph segments."""
        if not self.multi or not self._sync:
            return 0
        n = 0
        dead = self._dead or ()
        for b in self.buckets:
            live = [p for p in b.params if id(p) not in dead]
            if b.work is None and live and all(id(p) in done_ids and p.grad is not None for p in live):
                self._launch(b)
                n += 1
        return n

    def gradient_slice(self, p):
        """The arena slice that is (or will be) ``p.grad`` after a synchronised step, or None (no arena yet / a dead parameter)."""
        b = self._bucket_of.get(id(p))
        return b.views.get(id(p)) if b is not None else None

    def _on_grad(self, p):
        if not self._armed:
            return
        if self._dead is not None and id(p) in self._dead:
            return                      # a parameter believed dead got a gradient: picked up as a straggler at finish
        b = self._bucket_of[id(p)]
        b.pending -= 1
        if b.pending == 0 and b.work is None:
            self._launch(b)

    def _comm(self, dev):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        return self._comm_stream

    def _reduce(self, params):
        """Flatten the gradients of `params` and start their all-reduce -> (flat, work).  On a GPU both are issued with the
        comm stream current, after it has waited for the producing stream(s)."""
        grads = [p.grad for p in params]
        dev = grads[0].device
        if dev.type == "cuda":
            comm = self._comm(dev)
            comm.wait_stream(torch.cuda.current_stream(dev))
            from . import ops
            side = ops.wgrad_stream(dev)          # conv weight gradients may be produced on their own stream (ops._wgrad_side)
            if side is not None:
                comm.wait_stream(side)
            with torch.cuda.stream(comm):
                flat = torch.cat([g.reshape(-1).float() for g in grads])
                # RCCL averages inside the collective (ncclAvg); other backends sum and _write_back divides
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = torch.cat([g.reshape(-1).float() for g in grads])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return flat, work

    def _launch(self, b: _Bucket):
        b.included = [p for p in b.params if p.grad is not None]
        if not b.included:
            return
        if b.arena is not None and len(b.included) == len(b.live) and all(id(p) in b.views for p in b.included):
            (b.flat, b.work), b.in_place = self._reduce_in_place(b), True
        else:                               # no arena (first step / CPU-only parameters), or a live parameter without a gradient this step
            b.flat, b.work = self._reduce(b.included)

    def _reduce_in_place(self, b: _Bucket):
        """All-reduce of the bucket's arena itself.  Gradients the kernels did not write into their slices (the small vectors; anything
        autograd accumulated elsewhere) are moved in by one multi-tensor copy, and p.grad becomes the slice."""
        dev = b.arena.device
        src, dst = [], []
        b.held = []
        for p in b.live:
            v = b.views[id(p)]
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad.reshape(v.shape))
                dst.append(v)
                b.held.append((p, p.grad))
        op = dist.ReduceOp.AVG if (self._avg and dev.type == "cuda") else dist.ReduceOp.SUM
        if dev.type == "cuda":
            comm = self._comm(dev)
            comm.wait_stream(torch.cuda.current_stream(dev))
            from . import ops
            side = ops.wgrad_stream(dev)
            if side is not None:
                comm.wait_stream(side)
            with torch.cuda.stream(comm):
                if src:
                    torch._foreach_copy_(dst, src)
                    for t in src:
                        t.record_stream(comm)
                work = dist.all_reduce(b.arena, op=op, group=self.group, async_op=True)
        else:
            if src:
                torch._foreach_copy_(dst, src)
            work = dist.all_reduce(b.arena, op=op, group=self.group, async_op=True)
        for p in b.live:
            p.grad = b.views[id(p)]
        return b.arena, work

    def _finish_in_place(self, b: _Bucket, copy_back=False):
        dev = b.arena.device
        ctx = torch.cuda.stream(self._comm(dev)) if dev.type == "cuda" else contextlib.nullcontext()
        with ctx:
            b.work.wait()                   # with the comm stream current (see _write_back)
            if not (self._avg and dev.type == "cuda"):
                b.arena.div_(self.world)
            if copy_back and b.held:
                # the caller holds these gradient tensors (pre-allocated buffers, a graph's outputs, flat-optimiser views): the averaged
                # values go back INTO them and p.grad stays bound to them; gradients the kernels wrote into their slices have no
                # other tensor to go back to -- there p.grad is the slice
                torch._foreach_copy_([g for _, g in b.held], [b.views[id(p)].view_as(g) for p, g in b.held])
                for p, g in b.held:
                    p.grad = g
        b.held = []

    def _write_back(self, params, flat, work, copy_back=False):
        dev = flat.device
        ctx = torch.cuda.stream(self._comm(dev)) if dev.type == "cuda" else contextlib.nullcontext()
        with ctx:
            # wait() with the comm stream current: on the nccl/RCCL backend it orders the CURRENT stream behind the
            # collective, and the divide / copy-back below run on that same stream
            work.wait()
            if not (self._avg and dev.type == "cuda"):
                flat.div_(self.world)
            grads, views, off = [], [], 0
            for p in params:
                g = p.grad
                n = g.numel()
                v = flat[off:off + n].view_as(g)
                off += n
                if copy_back or g.dtype != flat.dtype or not g.is_contiguous():
                    grads.append(g)
                    views.append(v)
                else:
                    # no copy: the averaged gradient IS the slice of the bucket (0.15 ms of multi-tensor copies per step otherwise)
                    p.grad = v
            if grads:
                # one multi-tensor copy per bucket: ~330 separate copy launches cost ~1 ms of GPU time per step after the backward
                torch._foreach_copy_(grads, views)
        if dev.type == "cuda" and not copy_back:
            flat.record_stream(torch.cuda.current_stream(dev))      # its slices live on as p.grad, read by the caller's stream

    def finish_gradient_sync(self, copy_back=False):
        """Wait for the bucket all-reduces and hand the averaged gradients back: ``p.grad`` becomes a slice of its bucket (no copy), or
        with ``copy_back=True`` the existing ``p.grad`` tensors are overwritten (GraphedStep: its gradient tensors are the graph's).

        ALIASING (default, ``copy_back=False``): ``p.grad`` is REBOUND to a view of the bucket.  A reference taken to the previous
        ``p.grad`` tensor before this call (a user-held gradient list, a flat-gradient optimiser's views, pre-allocated gradient
        buffers) keeps the LOCAL, un-averaged gradient -- read gradients through ``p.grad`` after this call, or pass
        ``copy_back=True`` to have the averaged values written into the tensors you already hold (this holds for the zero-copy
        buckets too: a gradient that was NOT already its arena slice when the bucket was launched is copied back into the tensor
        that held it and ``p.grad`` stays bound to that tensor; a gradient the kernels wrote straight into the slice has no other
        home and stays the slice).  Each ``p.grad`` also keeps its
        whole bucket (``bucket_mb``) alive until it is released (``zero_grad(set_to_none=True)``).
        Buckets whose hooks did not all fire (parameters without gradients this step) are reduced here with what they have."""
        if not self.multi or not self._sync:
            self._armed = False
            return
        for b in self.buckets:
            if b.work is None:
                self._launch(b)
        late = []
        for b in self.buckets:
            if b.work is not None and b.in_place:
                self._finish_in_place(b, copy_back)
            elif b.work is not None:
                self._write_back(b.included, b.flat, b.work, copy_back)
            inc = {id(p) for p in b.included}
            late += [p for p in b.params if p.grad is not None and id(p) not in inc]
        if late:            # gradients that appeared after their bucket had been launched (a parameter that used to be dead)
            flat, work = self._reduce(late)
            self._write_back(late, flat, work, copy_back)
        if self._comm_stream is not None:
            torch.cuda.current_stream(self._comm_stream.device).wait_stream(self._comm_stream)
        # the same autograd graph runs on every rank, so this set is identical everywhere
        self._dead = {id(p) for b in self.buckets for p in b.params if p.grad is None}
        for b in self.buckets:
            b.flat, b.work, b.included, b.in_place = None, None, [], False
        self._armed = False

    sync_gradients = finish_gradient_sync
