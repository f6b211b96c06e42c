"""Child process of tests/test_gpu_distributed.py::test_product_api_two_ranks_one_gpu: one rank of a 2-rank `predict_from_folder` job
(RANK / WORLD_SIZE / MASTER_* from the environment, CF_DIST_BACKEND=gloo so that both ranks can share the box's single GPU).
argv: model_folder input_folder output_folder"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))

import torch  # noqa: E402


def main():
    model, inp, out = sys.argv[1:4]
    torch.cuda.set_device(0)
    from cineflow import predict
    # part_id 0 / num_parts 1 (the CLI defaults): the shard comes from RANK / WORLD_SIZE
    res = predict.predict_from_folder(model, inp, out, [0], False, 1, 2, None, 0, 1, True)
    print("rank %s wrote %s" % (os.environ.get("RANK"), sorted(res)))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
