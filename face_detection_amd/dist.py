"""Per-image sharding of the box pipeline over the GPUs of one node.

Every image is independent in both halves of the path (the reference processes one image
per call: utils/datasets.py:152, inference.py:37-60), so ranks own contiguous slices of the
batch, build anchors from the geometry locally and never exchange data inside the hot loop.
The only collective is optional: one all-gather of the fixed-size detection blocks so that
every rank ends up with all detections (RCCL over xGMI on GPUs; the same code runs on gloo
for CPU tests).  One process per GPU, launched by torch.distributed.run.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous [start, stop) of `n_items` owned by `rank`; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return (n_items * rank) // world_size, (n_items * (rank + 1)) // world_size


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (as set
    by torch.distributed.run) and binds this process to its GPU.  Returns
    (rank, world_size, local_rank); a single process without those variables is (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SSDBOX_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal
            backend = os.environ.get("SSDBOX_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def pack_detections(dets, keep_idx, count):
    """(b,limit,6) f32, (b,limit) i32, (b,) i32 -> one (b, limit*7 + 1) int32 block per image
    (float bits are carried as int32 so a single collective moves everything)."""
    b, limit = keep_idx.shape
    block = torch.empty((b, limit * 7 + 1), dtype=torch.int32, device=dets.device)
    block[:, :limit * 6] = dets.reshape(b, limit * 6).contiguous().view(torch.int32)
    block[:, limit * 6:limit * 7] = keep_idx
    block[:, limit * 7] = count
    return block


def unpack_detections(block, limit):
    b = block.shape[0]
    dets = block[:, :limit * 6].contiguous().view(torch.float32).reshape(b, limit, 6)
    keep = block[:, limit * 6:limit * 7].contiguous()
    count = block[:, limit * 7].contiguous()
    return dets, keep, count


def all_gather_detections(dets, keep_idx, count, global_batch, group=None):
    """Every rank contributes the detections of its shard_range slice of a `global_batch`
    image batch and receives all of them, in image order: one all_gather of equal-size
    blocks (short shards are padded by one image)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return dets, keep_idx, count
    rank = dist.get_rank(group)
    limit = keep_idx.shape[1]
    lo, hi = shard_range(global_batch, rank, world)
    if dets.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d images, its shard is [%d, %d)" % (rank, dets.shape[0], lo, hi))
    per_rank = -(-global_batch // world)
    block = torch.zeros((per_rank, limit * 7 + 1), dtype=torch.int32, device=dets.device)
    block[: hi - lo] = pack_detections(dets, keep_idx, count)
    device = dets.device
    if device.type == "cuda" and dist.get_backend(group) == "gloo":
        block = block.cpu()            # gloo gathers host memory; RCCL gathers in place over xGMI
    gathered = torch.empty((world * per_rank, limit * 7 + 1), dtype=torch.int32, device=block.device)
    dist.all_gather_into_tensor(gathered, block, group=group)
    gathered = gathered.to(device)
    parts = []
    for r in range(world):
        r_lo, r_hi = shard_range(global_batch, r, world)
        parts.append(gathered[r * per_rank: r * per_rank + (r_hi - r_lo)])
    return unpack_detections(torch.cat(parts, 0), limit)
