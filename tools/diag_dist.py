import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.parallel import Comm
comm = Comm.from_env("cuda")
dev = torch.device("cuda", 0)
t = torch.randn(5_200_000, device=dev)
for name, fn in (("all_reduce 20MB", lambda: comm.all_reduce_sum(t)), ("barrier", comm.barrier)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if comm.rank == 0: print(name, (time.perf_counter() - t0) / 3 * 1e3, "ms", flush=True)
# compute contention: a long kernel chain from both processes
a = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize(); comm.barrier()
t0 = time.perf_counter()
for _ in range(20): b = a @ a
torch.cuda.synchronize()
print(comm.rank, "20 matmuls", (time.perf_counter() - t0) * 1e3, "ms", flush=True)
comm.close()
