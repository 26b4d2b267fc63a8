"""The collectives bench.py and face_detection_amd.dist use (float64 MAX all-reduce, int32
all-gather, barrier) on RCCL with a single rank: what a one-GPU box can check of the N > 1 path."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
blk = torch.arange(7001 * 4, dtype=torch.int32, device="cuda").reshape(4, 7001)
out = torch.empty((4, 7001), dtype=torch.int32, device="cuda")
dist.all_gather_into_tensor(out, blk)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(t), bool((out == blk).all()))
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from face_detection_amd import dist as fdist
d = torch.rand(4, 1000, 6, device="cuda"); k = torch.randint(0, 100, (4, 1000), dtype=torch.int32, device="cuda"); c = torch.randint(0, 1000, (4,), dtype=torch.int32, device="cuda")
a, b, cc = fdist.all_gather_detections(d, k, c, 4)
print("gather ok", bool((a == d).all() and (b == k).all() and (cc == c).all()))
dist.destroy_process_group()
