"""RCCL smoke test on the GPU box: the multi-GPU host path (weight broadcast as one flat buffer + MAX/SUM reductions)
through the real "nccl" (= RCCL) backend with a single rank.  N > 1 needs a multi-GPU node (driver's scaling run); the
partition / broadcast logic for world_size 2 is covered on CPU by tests/test_distributed_cpu.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_broadcast_and_reductions(dev):
    import torch.distributed as dist
    from cineflow import parallel
    from cineflow.models import Generic_UNet
    from cineflow.weights import seeded_state_dict
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    old = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        net = Generic_UNet(1, 8, 4, 3)
        shapes = net.state_shapes()
        sd = seeded_state_dict(shapes, 3)
        got = parallel.broadcast_state_dict(sd, shapes, dev)
        flat = torch.cat([got[k].reshape(-1) for k in sorted(shapes)])
        dist.broadcast(flat, src=0)          # the collective itself, on the GPU buffer
        for k in shapes:
            assert got[k].is_cuda and torch.equal(got[k].cpu(), sd[k])
        net.load_state_dict(got, dev)
        assert parallel.max_over_ranks(1.25, dev) == 1.25 and parallel.sum_over_ranks(3.0, dev) == 3.0
        parallel.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
