"""The engine's communicator over torch's RCCL process group with ONE rank (all a 1-GPU box holds): the library's own RCCL
communicator (default) or, with BDF_COMM_FORCE_STAGED=1, the agreed fall-back -- torch.distributed's all-gather behind the
library's host transport.  An in-place exchange of three chunks must leave the buffer as it was; prints the transport."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
import bdf_amd as B
from bdf_amd._lib import check, lib
from bdf_amd.engine import Comm, Context, _ptr

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 90))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = Context(0, 5)
comm = Comm(ctx, 0, 1)
D, chunks, cmax = 8, 3, 50
x = ctx.zeros(chunks * cmax, D)
with torch.cuda.stream(ctx.stream):
    x.copy_(torch.arange(chunks * cmax * D, dtype=torch.float64, device=x.device).reshape(chunks * cmax, D))
ref = x.clone()
for c in range(chunks):
    check(lib().bdf_allgather_rows(ctx.handle, comm.handle, D, chunks * cmax, _ptr(x), c, chunks))
check(lib().bdf_allgather_join(ctx.handle, comm.handle))
s = ctx.tensor([1.5, -2.0])
check(lib().bdf_sum_ranks(ctx.handle, comm.handle, _ptr(s), 2))           # one rank: unchanged
ctx.sync()
assert torch.equal(x, ref) and s.cpu().tolist() == [1.5, -2.0]
print("transport:", comm.transport, flush=True)
comm.close()
ctx.close()
dist.destroy_process_group()
