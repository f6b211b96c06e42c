"""Peak device memory of one bench step (BASELINE config 4) at 64 and 128 slices per step."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
dev = torch.device("cuda:0")
for B in (64, 128):
    torch.cuda.reset_peak_memory_stats()
    fnet, snet = bench.load_nets(bench.make_nets("video"), dev, 1234, 1, 0)
    frames = bench.synthetic_cine(B, 30, 256, 1).to(dev)
    out = bench.run_step(fnet, snet, frames, None)
    torch.cuda.synchronize()
    print("B", B, "peak allocated GiB %.1f  reserved GiB %.1f" % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30), flush=True)
    del out, frames, fnet, snet
    torch.cuda.empty_cache()
