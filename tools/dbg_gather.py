"""Host cost of the boundary gather's pieces (one rank, RCCL initialised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from pypore_amd import dist as pdist
b = torch.arange(9735, dtype=torch.int32, device="cuda")
bg = pdist.BoundaryGather(32768, b.device, b.dtype)
def t(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def both():
    k = bg.submit(b); bg.result(k)
print("submit+result  %.1f us" % t(both))
def sub_only():
    k = bg.submit(b); s = bg.slots[k % 2]; s["work"].wait(); s["work"] = None
print("submit+wait    %.1f us" % t(sub_only))
s = bg.slots[0]
print("fill           %.1f us" % t(lambda: s["send"].__setitem__(0, 5)))
print("copy           %.1f us" % t(lambda: s["send"].__setitem__(slice(1, 9736), b)))
print("all_gather     %.1f us" % t(lambda: dist.all_gather_into_tensor(s["recv"], s["send"])))
print("counts to host %.1f us" % t(lambda: s["recv"].view(1, -1)[:, 0].cpu().tolist()))
print("varlen         %.1f us" % t(lambda: pdist.gather_varlen(b)))
dist.destroy_process_group()
