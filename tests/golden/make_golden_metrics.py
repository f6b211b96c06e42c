#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by RUNNING THE REFERENCE's metric functions in the build container and pin oracle/metrics.py.

    cd /tmp && python /root/repo/tests/golden/make_golden_metrics.py

medpy, kornia and pystrum are absent from the image.  `nnunet.evaluation.metrics.metric` (medpy.metric) is therefore the oracle's
restatement of medpy.metric.binary's hd / hd95 / asd / assd, and `pystrum.pynd.ndutils.volsize2ndgrid` a two-line meshgrid: the pins
cover the reference's own ConfusionMatrix, ratios, NaN rules and jacobian_determinant; the medpy bodies and kornia's
spatial_gradient3d stay "parity unpinned" (oracle/metrics.py).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import metrics as OM  # noqa: E402

REPORT = []


def pin(name, ref, ora, tol=0.0):
    ref, ora = np.asarray(ref, dtype=np.float64), np.asarray(ora, dtype=np.float64)
    assert ref.shape == ora.shape, (name, ref.shape, ora.shape)
    both_nan = np.isnan(ref) & np.isnan(ora)
    d = float(np.abs(np.where(both_nan, 0.0, ref - ora)).max()) if ref.size else 0.0
    assert not np.isnan(d), name
    REPORT.append((name, d, tol))
    print("  oracle vs reference %-46s max|diff| = %.3e (tol %.1e)" % (name, d, tol))
    assert d <= tol, name


def blobs(seed, shape):
    """two overlapping label maps (0..3) made of shifted ellipsoids"""
    rng = np.random.default_rng(seed)
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    out = []
    for shift in (0.0, 1.7):
        lab = np.zeros(shape, np.uint8)
        for c, rad in ((1, 0.42), (2, 0.30), (3, 0.17)):
            cen = [s / 2 + shift * (i + 1) / 2 + rng.normal() for i, s in enumerate(shape)]
            r2 = sum(((g - ce) / (rad * s + 1e-9)) ** 2 for g, ce, s in zip(grids, cen, shape))
            lab[r2 <= 1.0] = c
        out.append(lab)
    return out


def main():
    import nnunet.evaluation.metrics as ref_m
    import nnunet.compute_jacobian as ref_j

    ref_m.metric = OM.medpy_binary
    ref_j.nd = types.SimpleNamespace(volsize2ndgrid=lambda volshape: np.meshgrid(*[np.arange(e) for e in volshape], indexing="ij"))

    fx = {}
    names = ("dice", "jaccard", "precision", "sensitivity", "specificity", "accuracy")
    surf = ("hausdorff_distance", "hausdorff_distance_95", "avg_surface_distance", "avg_surface_distance_symmetric")
    for tag, shape, spacing in (("3d", (9, 40, 36), (10.0, 1.25, 1.5)), ("2d", (48, 44), (1.4, 1.1))):
        test, gt = blobs(3 if tag == "3d" else 4, shape)
        fx[tag + "_test"], fx[tag + "_gt"], fx[tag + "_spacing"] = test, gt, np.array(spacing)
        for c in (1, 2, 3):
            a, b = test == c, gt == c
            vals = [getattr(ref_m, n)(a, b) for n in names]
            pin("%s class %d ratios" % (tag, c), vals, [getattr(OM, n)(a, b) for n in names])
            sv = [getattr(ref_m, n)(a, b, voxel_spacing=spacing) for n in surf]
            pin("%s class %d surface distances" % (tag, c), sv, [getattr(OM, n)(a, b, voxel_spacing=spacing) for n in surf])
            fx["%s_c%d_ratios" % (tag, c)], fx["%s_c%d_surface" % (tag, c)] = np.array(vals), np.array(sv)
    # the NaN rules: empty / full test or reference
    e, f = np.zeros((6, 7), bool), np.ones((6, 7), bool)
    h = e.copy()
    h[2:4, 1:5] = True
    cases = {"empty_empty": (e, e), "empty_ref": (h, e), "empty_test": (e, h), "full_test": (f, h), "full_ref": (h, f)}
    for k, (a, b) in cases.items():
        vals = [getattr(ref_m, n)(a, b) for n in names] + [getattr(ref_m, n)(a, b) for n in surf]
        pin("NaN rules " + k, vals, [getattr(OM, n)(a, b) for n in names] + [getattr(OM, n)(a, b) for n in surf])
        fx["nan_" + k] = np.array(vals)
    pin("nan_for_nonexisting=False", [ref_m.dice(e, e, nan_for_nonexisting=False), ref_m.hausdorff_distance(e, h, nan_for_nonexisting=False)],
        [OM.dice(e, e, False), OM.hausdorff_distance(e, h, False)])

    # compute_jacobian.py: determinant of a smooth 2-D field and the per-structure statistics (script lines restated in the oracle)
    rng = np.random.default_rng(9)
    from scipy.ndimage import gaussian_filter
    T, H, W = 5, 48, 44
    flow = np.stack([gaussian_filter(rng.normal(size=(T, H, W)), (1.0, 3.0, 3.0)) for _ in range(2)], -1).astype(np.float32) * 25
    gt2 = blobs(5, (H, W))[1]
    jac = np.stack([ref_j.jacobian_determinant(flow[t]) for t in range(T)])
    pin("jacobian_determinant", jac, np.stack([OM.jacobian_determinant(flow[t]) for t in range(T)]))
    assert (jac < 0).any(), "the fixture should contain folded voxels"
    st = OM.jacobian_frame_stats(flow[2], gt2)
    keys = sorted(st)
    # the script's own arithmetic (compute_jacobian.py:160-186) replayed on the reference's determinant
    cur = {}
    for i, k in enumerate(("RV", "MYO", "LV"), 1):
        cj = jac[2][gt2 == i]
        cur["abs(Mean jacobian - 1)_" + k] = abs(cj.mean() - 1)
        cur["total_" + k] = float(cj.size)
        cur["negative_" + k] = float((cj < 0).sum())
        cur["negative_%_" + k] = (cur["negative_" + k] / cur["total_" + k]) * 100
    cur["abs(Mean jacobian - 1)_average"] = (cur["abs(Mean jacobian - 1)_LV"] + cur["abs(Mean jacobian - 1)_RV"] + cur["abs(Mean jacobian - 1)_MYO"]) / 3
    cur["negative_%_average"] = (cur["negative_%_LV"] + cur["negative_%_RV"] + cur["negative_%_MYO"]) / 3
    cur["abs(Mean jacobian - 1)"] = abs(jac[2].mean() - 1)
    cur["total"], cur["negative"] = float(jac[2].size), float((jac[2] < 0).sum())
    cur["negative_%"] = (cur["negative"] / cur["total"]) * 100
    pin("jacobian frame statistics", [cur[k] for k in keys], [st[k] for k in keys], 1e-12)
    tg, sg = OM.gradient_means(flow)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), flow=flow, gt2=gt2, jac=jac, stats_keys=np.array(keys), stats=np.array([cur[k] for k in keys]),
                        temporal_gradient=tg, spatial_gradient=sg, **fx)
    print("wrote metrics.npz %.1f KiB" % (os.path.getsize(os.path.join(HERE, "metrics.npz")) / 1024))
    print("\nall %d metric pins within tolerance" % len(REPORT))
    with open(os.path.join(HERE, "PIN_REPORT_metrics.txt"), "w") as f:
        f.write("oracle/metrics.py vs the reference's nnunet/evaluation/metrics.py and nnunet/compute_jacobian.py (make_golden_metrics.py, build container);\n")
        f.write("medpy.metric.binary bodies and kornia.spatial_gradient3d (absent packages) are the oracle's restatements -- parity unpinned for those\n")
        for n, d, t in REPORT:
            f.write("%-56s max|diff| %.3e  tol %.1e\n" % (n, d, t))


if __name__ == "__main__":
    main()
