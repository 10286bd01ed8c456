# two ranks on ONE GPU: does RCCL accept it?  (expected: no — "duplicate GPU"; bench.py uses gloo when ranks share a device)
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, ".")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from ams_amd.dist import RcclComm
try:
    c = RcclComm(rank, world, torch.device("cuda:0"))
    t = torch.ones(4, device="cuda:0") * (rank + 1)
    c.all_reduce(t); torch.cuda.synchronize()
    print("rank", rank, "RCCL on a shared GPU works:", t.tolist(), flush=True)
except Exception as e:
    print("rank", rank, "RCCL on a shared GPU refused:", str(e)[:200], flush=True)
dist.destroy_process_group()
