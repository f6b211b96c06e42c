"""world_size-2 gloo test of the multi-GPU host logic (no GPU): weight broadcast as one flat buffer, the reference's
[part_id::num_parts] patient partition, and the max-over-ranks timing reduction used by bench.py."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cardiac-segmentation-optical-flow_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cineflow import parallel
    from cineflow.weights import seeded_state_dict
    r, w, _ = parallel.init_from_env(backend="gloo")
    shapes = {"a.conv1.weight": (4, 3, 3, 3), "a.norm1.weight": (4,), "a.norm1.bias": (4,), "b.linear1.weight": (8, 4)}
    sd = seeded_state_dict(shapes, seed=5) if r == 0 else None  # only rank 0 "reads the checkpoint"
    got = parallel.broadcast_state_dict(sd, shapes, torch.device("cpu"))
    want = seeded_state_dict(shapes, seed=5)
    ok = all(torch.equal(got[k], want[k]) for k in shapes)
    patients = list(range(7))
    mine = parallel.shard(patients, r, w)
    tmax = parallel.max_over_ranks(1.0 + r, torch.device("cpu"))
    tsum = parallel.sum_over_ranks(len(mine), torch.device("cpu"))
    parallel.barrier()
    q.put((r, ok, mine, tmax, tsum))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "broadcast weights differ from rank 0's"
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert res[0][3] == res[1][3] == 2.0
    assert res[0][4] == res[1][4] == 7.0


def _run_bench(*args):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), env=env, capture_output=True, text=True, timeout=600)


def test_bench_self_launch_dry_run_two_ranks():
    """`python bench.py --gpus 2` starts its own ranks (the driver's contract): dry-run mode goes through the launcher, the
    rendezvous, the flat weight broadcast of the full-size networks, the barrier and the max-over-ranks reduction on gloo and
    prints exactly ONE JSON line from rank 0."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True and d["scaling"] == "weak"
    assert d["weights_identical_on_all_ranks"] is True and d["ranks_reporting"] == 2
    assert d["ms_per_step"] >= 20.0          # rank 1 sleeps 20 ms per step: the MAX over ranks is reported


def test_bench_self_launch_propagates_rank_failure():
    """a rank that fails (here: no GPU in the build container) makes the launcher exit non-zero and print no JSON line"""
    r = _run_bench("--gpus", "2", "--steps", "1", "--no-raft")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_warp_variant_inputs_and_dry_run():
    """BASELINE config 2 variant: the synthetic frame pairs are deterministic per seed (ranks seed by rank, so shards differ), labels hold
    the four classes, the displacement is a few pixels and smooth; the launcher path runs for this variant too."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    f0, i0, l0 = bench.synthetic_pairs(3, 64, 11)
    f1, i1, l1 = bench.synthetic_pairs(3, 64, 11)
    f2, _, _ = bench.synthetic_pairs(3, 64, 12)
    assert torch.equal(f0, f1) and torch.equal(i0, i1) and torch.equal(l0, l1) and not torch.equal(f0, f2)
    assert f0.shape == (3, 2, 64, 64) and i0.shape == (3, 1, 64, 64) and l0.shape == (3, 64, 64) and l0.dtype == torch.uint8
    assert sorted(l0.unique().tolist()) == [0, 1, 2, 3] and 0.5 < float(f0.abs().max()) < 20.0
    assert float((f0[:, :, 1:] - f0[:, :, :-1]).abs().max()) < 5.0          # smooth: a 16x16 field stretched to 64x64 (four times gentler at 256)
    r = _run_bench("--gpus", "2", "--dry-run", "--variant", "warp", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert d["config"]["variant"] == "warp" and d["ranks_reporting"] == 2


def test_bench_dry_run_eight_ranks():
    """the driver's N = 8 launch shape rehearsed on gloo: eight fresh children, one rendezvous, the flat broadcast of the full-size weights to
    seven receivers, barrier, MAX over eight ranks (rank 7 sleeps 80 ms per step), one JSON line"""
    import json
    r = _run_bench("--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_reporting"] == 8 and d["weights_identical_on_all_ranks"] is True
    assert d["ms_per_step"] >= 80.0
    assert sum("weight broadcast" in l for l in r.stderr.splitlines()) >= 8          # every rank reports its broadcast time on stderr


def test_bench_successive_and_api_variants_dry_run():
    import json
    for variant in ("successive", "api"):
        r = _run_bench("--gpus", "2", "--dry-run", "--variant", variant, "--steps", "1", "--warmup", "0")
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
        assert d["config"]["variant"] == variant and d["weights_identical_on_all_ranks"] is True


def _api_worker(rank, world, port, q, folders):
    """cineflow.predict.load_model_and_checkpoint_files in a 2-rank job: rank 1's model folder holds plans.json ONLY"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cardiac-segmentation-optical-flow_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cineflow import parallel, predict
    parallel.init_from_env(backend="gloo")
    trainer, params = predict.load_model_and_checkpoint_files(folders[rank], folds=None, mixed_precision=True, device=torch.device("cpu"))
    sums = {}
    for part in ("seg_state_dict", "flow_state_dict"):
        sums[part] = (len(params[0][part]), float(sum(float(v.double().sum()) for v in params[0][part].values())))
    parallel.barrier()
    q.put((rank, len(params), sums, bool(trainer.mixed_precision)))
    torch.distributed.destroy_process_group()


def test_product_api_weight_broadcast_two_ranks(tmp_path):
    """the weight broadcast sits in the product API (cineflow.predict.load_model_and_checkpoint_files), in front of the
    patient loop of the --part_id/--num_parts path (reference contract: nnunet/inference/predict.py:743,806-821).  Rank 0 reads
    fold_0/model_final_checkpoint.model; rank 1 has plans.json only and must end up with the same tensors."""
    import json
    import shutil
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cardiac-segmentation-optical-flow_amd"))
    from cineflow import predict
    from cineflow.models import Generic_UNet, SegFlowGaussian
    from cineflow.weights import seeded_state_dict
    reduced = dict(in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=32, bottleneck_heads=4, dim_feedforward=48)
    plans = predict.default_plans(image_size=64, crop_size=64, flow_variant="video", seg_base=8, seg_pool=3, reduced=reduced)
    seg = Generic_UNet(1, 8, 4, 3)
    flow = SegFlowGaussian(image_size=64, motion_appearance=False, **reduced)
    seg_sd = seeded_state_dict({k: v for k, v in seg.state_shapes().items()}, seed=3)
    flow_sd = seeded_state_dict({k: v for k, v in flow.state_shapes().items() if not k.endswith("grid")}, seed=4)
    f0, f1 = str(tmp_path / "rank0"), str(tmp_path / "rank1")
    predict.save_model_folder(f0, seg, flow, plans, seg_sd=seg_sd, flow_sd=flow_sd)
    os.makedirs(f1)
    shutil.copy(os.path.join(f0, "plans.json"), f1)
    assert sorted(os.listdir(f1)) == ["plans.json"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_api_worker, args=(r, 2, port, q, [f0, f1])) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_seg = float(sum(float(v.double().sum()) for v in seg_sd.values()))
    want_flow = float(sum(float(v.double().sum()) for v in flow_sd.values()))
    for rank, nfold, sums, mp_flag in res:
        assert nfold == 1 and mp_flag is True
        assert sums["seg_state_dict"][0] == len(seg_sd) and abs(sums["seg_state_dict"][1] - want_seg) < 1e-6 * (1 + abs(want_seg))
        assert sums["flow_state_dict"][0] == len(flow_sd) and abs(sums["flow_state_dict"][1] - want_flow) < 1e-6 * (1 + abs(want_flow))
    assert json.load(open(os.path.join(f1, "plans.json")))["num_classes"] == 4
