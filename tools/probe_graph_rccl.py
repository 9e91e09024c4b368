"""Round-5 probe (VERDICT r4 #6b): can an RCCL all_reduce be captured into a HIP graph on this stack (group of one rank, side stream forked
from the capture stream)?  Result on MI355X / ROCm 7.2 / torch 2.10: the process dumps core inside the capture -- collectives stay outside
the graphs (GraphedStep replays segments and launches the bucket all-reduces between them).  python tools/probe_graph_rccl.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros_like(x)
comm = torch.cuda.Stream()
# warm-up (eager) so that the communicator exists
dist.all_reduce(x); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.mul_(2.0)
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            w = dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True)
            w.wait()
        y.add_(1.0)                       # independent work on the capture stream: may overlap the collective
        torch.cuda.current_stream().wait_stream(comm)
        y.add_(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured all_reduce OK: x[0] =", float(x[0]), "y[0] =", float(y[0]))
except Exception as e:
    print("capture FAILED:", repr(e)[:300])
dist.destroy_process_group()
