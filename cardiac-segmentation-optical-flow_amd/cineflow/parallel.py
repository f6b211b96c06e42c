"""Patient-parallel multi-GPU support: one process per GPU, no data-path collective.

The reference's multi-GPU inference is "run N OS processes with --part_id/--num_parts"
(nnunet/inference/predict.py:806-821, slice `lists[part_id::num_parts]` at :743).  The build keeps exactly that
partition (rank = part_id, world_size = num_parts) and adds ONE collective the reference does not have: a broadcast
of the flattened weights from rank 0 at start-up (RCCL over xGMI when the backend is "nccl"), so only rank 0 touches
the checkpoint files.  After that ranks never communicate, except for the optional scalar reduction of the timing
report.  The same code runs on gloo/CPU tensors for the world_size-2 tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world_size, local_rank).
    With WORLD_SIZE unset or 1 nothing is initialised."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # CF_DIST_BACKEND=gloo: several ranks on ONE GPU (the 2-rank GPU test of the product API on a 1-GPU box; RCCL refuses two ranks
            # on one device)
            backend = os.environ.get("CF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        # CF_DIST_DEVICE_ID=0 falls back to the lazy communicator initialisation of rounds 1-2 (no device_id): the eager path has only
        # ever run with one rank on hardware, so a failure at scale can be bisected in a single run
        if backend == "nccl" and os.environ.get("CF_DIST_DEVICE_ID", "1") != "0":
            # bind this rank to ITS GPU before the process group exists and tell the group so: without device_id the first collective opens
            # a lazy context on device 0 from every rank (memory and time lost on GPU 0 of an 8-rank job).  Every GPU stays visible to every
            # rank -- RCCL's xGMI peer paths need that -- so no *_VISIBLE_DEVICES masking here.
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


LAST_BROADCAST = {}      # {'seconds', 'bytes'} of the last flat weight broadcast on this rank (bench.py reports it on stderr)


def shard(items, part_id, num_parts):
    """The reference's partition: items[part_id::num_parts] (predict.py:743)."""
    assert 0 <= part_id < num_parts
    return list(items)[part_id::num_parts]


def broadcast_state_dict(sd, shapes, device, src=0):
    """Broadcast a weight dict from rank `src` as ONE flat fp32 buffer.

    sd     : {name: tensor} on rank `src` (ignored elsewhere, may be None)
    shapes : {name: shape} known on every rank (Module.state_shapes())
    Returns {name: tensor on `device`} on every rank.  One message instead of hundreds: a flat rank-0 -> all
    broadcast of ~0.1-0.25 GB uses all xGMI links in parallel; per-tensor broadcasts would be launch-latency bound.
    """
    names = sorted(shapes)
    sizes = [int(torch.Size(shapes[n]).numel()) for n in names]
    total = sum(sizes)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        off = 0
        for n, s in zip(names, sizes):
            flat[off:off + s].copy_(sd[n].reshape(-1).to(torch.float32))
            off += s
    if world > 1:
        import time
        t0 = time.perf_counter()
        dist.broadcast(flat, src=src)
        if flat.is_cuda:
            torch.cuda.synchronize(flat.device)
        LAST_BROADCAST.update(seconds=time.perf_counter() - t0, bytes=total * 4)
    out, off = {}, 0
    for n, s in zip(names, sizes):
        out[n] = flat[off:off + s].view(shapes[n])
        off += s
    return out


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (the bench contract's max-over-ranks step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
