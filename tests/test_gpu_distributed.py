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


def test_product_api_two_ranks_one_gpu(dev, tmp_path):
    """predict_from_folder as a 2-rank job (one process per rank, both on this box's GPU, gloo carrying the device buffers): rank 1's model
    folder holds plans.json ONLY -- its weights arrive through the flat broadcast inside load_model_and_checkpoint_files -- the shard is
    patients[rank::world] (predict.py:743), and every output equals the single-process run's (label maps identical, flows to 2e-5 px)."""
    import shutil
    import subprocess
    import sys
    import numpy as np
    from cineflow import predict as P
    from cineflow.models import SegFlowGaussian, Generic_UNet
    from cineflow.nifti import read_nifti, write_nifti
    from cineflow.weights import seeded_state_dict
    red = dict(in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=32, bottleneck_heads=4, dim_feedforward=48)
    plans = P.default_plans(image_size=64, crop_size=64, flow_variant="video", seg_base=8, seg_pool=3, reduced=red)
    seg = Generic_UNet(1, 8, 4, 3)
    flow = SegFlowGaussian(image_size=64, motion_appearance=False, **red)
    sd_s = seeded_state_dict({k: v for k, v in seg.state_shapes().items()}, 10)
    sd_f = seeded_state_dict({k: v for k, v in flow.state_shapes().items() if not k.endswith("grid")}, 11)
    m0, m1 = str(tmp_path / "model_rank0"), str(tmp_path / "model_rank1")
    P.save_model_folder(m0, seg, flow, plans, fold=0, seg_sd=sd_s, flow_sd=sd_f)
    os.makedirs(m1)
    shutil.copy(os.path.join(m0, "plans.json"), m1)
    inp = tmp_path / "in"
    g = torch.Generator().manual_seed(5)
    T, Z, Y, X = 3, 2, 60, 56
    pats = ["patient%03d" % i for i in range(1, 4)]
    for pat in pats:
        (inp / pat).mkdir(parents=True)
        for t in range(T):
            vol = torch.randn(Z, Y, X, generator=g).numpy().astype(np.float32) * 40 + 100
            write_nifti(str(inp / pat / ("%s_frame%02d_0000.nii.gz" % (pat, t))), vol, (1.5, 1.5, 8.0), (0, 0, 0))
    single = str(tmp_path / "out_single")
    P.clear_model_cache()
    P.predict_from_folder(m0, str(inp), single, [0], False, 1, 2, None, 0, 1, True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = [str(tmp_path / "out_rank0"), str(tmp_path / "out_rank1")]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_api_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CF_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, worker, [m0, m1][r], str(inp), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    shards = {0: pats[0::2], 1: pats[1::2]}
    for r in range(2):
        assert sorted(d for d in os.listdir(outs[r]) if d.startswith("patient")) == shards[r]
        for pat in shards[r]:
            for t in range(T):
                case = "%s_frame%02d" % (pat, t)
                for sub in ("Segmentation", "Registered"):
                    a, _ = read_nifti(os.path.join(outs[r], pat, sub, case + ".nii.gz"))
                    b, _ = read_nifti(os.path.join(single, pat, sub, case + ".nii.gz"))
                    assert np.array_equal(a, b), (r, pat, sub, t)
                fa = np.load(os.path.join(outs[r], pat, "Flow", case + ".npz"))["flow"]
                fb = np.load(os.path.join(single, pat, "Flow", case + ".npz"))["flow"]
                # (the single-process run batches all three patients into one device call, a rank one or two: kernel shapes are chosen per
                # launch size, so the flows agree to fp32 summation noise, not bitwise; the label maps above are identical)
                assert fa.shape == fb.shape and float(np.abs(fa - fb).max()) <= 2e-5, (r, pat, t)
