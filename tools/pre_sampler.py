#!/usr/bin/env python3
"""Where do the preprocessing threads of the file-level API spend their time?  N threads preprocess synthetic frames (gzip NIfTI in tmpfs) on
HIP streams of their own while a sampler thread records every thread's innermost frames every 2 ms (sys._current_frames).
    python tools/pre_sampler.py [threads=8] [frames per thread=24]"""
import collections
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cardiac-segmentation-optical-flow_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cineflow import predict as P  # noqa: E402
from cineflow.nifti import write_nifti  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="cf_pre_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    g = torch.Generator().manual_seed(0)
    files = []
    for i in range(nthreads * per):
        a = (torch.rand(8, 256, 256, generator=g).numpy() * 60.0 + 200.0).astype(np.float32)
        f = os.path.join(root, "f%04d_0000.nii.gz" % i)
        write_nifti(f, a, (1.25, 1.25, 10.0))
        files.append([f])
    trainer = P.CineTrainer(P.default_plans(image_size=256, crop_size=256, flow_variant="video", seg_base=8, seg_pool=3, reduced={"reduced": True}) if False
                            else P.default_plans(image_size=256, crop_size=256), dev)
    trainer.preprocess_patient(files[0])           # warm
    torch.cuda.synchronize()

    def one_run(n):
        hist = collections.Counter()
        stop = [False]
        workers = []

        def work(k):
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for j in range(per):
                    trainer.preprocess_patient(files[k * per + j])
                    st.synchronize()

        def sample():
            ids = None
            while not stop[0]:
                fr = sys._current_frames()
                ids = ids or {t.ident for t in workers}
                for tid, f in fr.items():
                    if tid not in ids:
                        continue
                    chain = []
                    while f is not None and len(chain) < 3:
                        chain.append("%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, f.f_code.co_name))
                        f = f.f_back
                    hist[" <- ".join(chain)] += 1
                time.sleep(0.002)

        workers.extend(threading.Thread(target=work, args=(k,)) for k in range(n))
        t0 = time.perf_counter()
        for w in workers:
            w.start()
        s = threading.Thread(target=sample)
        s.start()
        for w in workers:
            w.join()
        dt = time.perf_counter() - t0
        stop[0] = True
        s.join()
        tot = sum(hist.values())
        print("== %d threads x %d frames: %.2f s wall, %.1f ms per frame per thread, %.1f ms wall per frame" % (n, per, dt, dt / per * 1e3, dt / (n * per) * 1e3))
        for k, v in hist.most_common(14):
            print("   %5.1f %%  %s" % (100.0 * v / max(tot, 1), k[:200]))

    one_run(1)
    one_run(nthreads)
    # the same frames through multiprocessing.pool.ThreadPool.apply_async, as cineflow.predict._predict_patients submits them
    from multiprocessing.pool import ThreadPool
    tls = threading.local()

    def pre_one(fl):
        if not hasattr(tls, "st"):
            torch.cuda.set_device(dev)
            tls.st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(tls.st):
            r = trainer.preprocess_patient(fl)
            tls.st.synchronize()
        return r
    for keep in (False, True):
        pool = ThreadPool(nthreads)
        t0 = time.perf_counter()
        res = [pool.apply_async(pre_one, (f,)) for f in files]
        out = []
        for a in res:
            r = a.get()
            if keep:
                out.append(r)
        dt = time.perf_counter() - t0
        pool.close()
        pool.join()
        print("== ThreadPool(%d).apply_async x %d frames, results %s: %.2f s wall, %.1f ms wall per frame" % (nthreads, len(files), "kept" if keep else "dropped", dt, dt / len(files) * 1e3))
        del out
    if len(sys.argv) > 3:           # third argument: also after the full-size networks have been loaded and have run one batch in this process
        import bench
        from cineflow.inference import predict_cine_slices
        fnet, snet = bench.load_nets(bench.make_nets("video"), dev, 1234, 1, 0)
        frames = bench.synthetic_cine(int(sys.argv[3]), 30, 256, 1234).to(dev)
        ed = bench.ring_labels(int(sys.argv[3]), 256).to(dev)
        bench.run_step(fnet, snet, frames, ed)
        torch.cuda.synchronize()
        print("-- after loading the networks and one batch of %s slices (allocator: %.1f GiB reserved)" % (sys.argv[3], torch.cuda.memory_reserved() / 2**30))
        one_run(nthreads)
        del frames, ed
        torch.cuda.empty_cache()
        print("-- after torch.cuda.empty_cache() (%.1f GiB reserved)" % (torch.cuda.memory_reserved() / 2**30))
        one_run(nthreads)
finally:
    shutil.rmtree(root, ignore_errors=True)
