#!/usr/bin/env python3
"""Generate tests/golden/strain.npz by RUNNING THE REFERENCE in the build container and pin oracle/strain.py.

    cd /tmp && python /root/repo/tests/golden/make_golden_strain.py

Imported in place from /root/reference: `nnunet.network_architecture.integration.SpatialTransformerContour` (the contour sampler of
get_strain.py) and get_strain.py's `curvature` / `smoothness_measure` (the module needs cv2, nibabel, skimage, matplotlib: inert import
stubs, tests/golden/_ref_import.py; its `fourcc = cv.VideoWriter_fourcc(*'mp4v')` at import time instantiates a stub object and nothing
else).  The strain curves and tracking errors of get_strain.py read .mat / NIfTI / pickle files inside the same functions and cannot be
called on arrays: they are restated (oracle/strain.py) on top of the pinned sampler, and stored here as oracle outputs for the GPU box.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import _ref_import  # noqa: E402

_ref_import.install()
sys.path.insert(0, os.path.join(_ref_import.REFERENCE_ROOT, "nnunet"))     # get_strain.py imports `network_architecture.integration`

from oracle import strain as OS  # noqa: E402

REPORT = []


def pin(name, ref, ora, tol=0.0):
    ref, ora = np.asarray(ref, dtype=np.float64), np.asarray(ora, dtype=np.float64)
    assert ref.shape == ora.shape, (name, ref.shape, ora.shape)
    d = float(np.abs(ref - ora).max())
    REPORT.append((name, d, tol))
    print("  oracle vs reference %-46s max|diff| = %.3e (tol %.1e)" % (name, d, tol))
    assert d <= tol, name


def main():
    from nnunet.network_architecture.integration import SpatialTransformerContour
    out = {}
    g = torch.Generator().manual_seed(77)
    # ---- the contour sampler on a non-square field (A = 40, B = 56), points partly outside
    A, Bb, P = 40, 56, 37
    field = torch.randn(2, 2, A, Bb, generator=g) * 3
    pts = torch.stack([torch.rand(2, 1, P, generator=g) * (Bb + 6) - 3, torch.rand(2, 1, P, generator=g) * (A + 6) - 3], dim=1)   # ch 0 along the last axis
    ref = SpatialTransformerContour(size=(A, Bb))(torch.clone(pts), field)
    pin("SpatialTransformerContour", ref, OS.spatial_transformer_contour(pts, field, (A, Bb)))
    out.update(stc_field=field.numpy(), stc_pts=pts.numpy(), stc_out=ref.numpy())
    # ---- curvature / smoothness of get_strain.py
    try:
        import importlib
        gs = importlib.import_module("get_strain")
        x = np.arange(25, dtype=np.float64)
        y = np.sin(x / 3.0) + 0.1 * np.cos(x * 1.7)
        pin("get_strain.curvature", gs.curvature(x, y), OS.curvature(x, y))
        pin("get_strain.smoothness_measure", gs.smoothness_measure(x, y), OS.smoothness_measure(x, y))
        out.update(curv_x=x, curv_y=y, curv_out=gs.curvature(x, y), smooth_out=np.array(gs.smoothness_measure(x, y)))
    except Exception as e:  # pragma: no cover
        print("  get_strain.py not importable here (%s: %s): curvature / smoothness_measure stay parity unpinned" % (type(e).__name__, e))
    # ---- oracle outputs of the restated strain chain on a synthetic contracting ring (inputs + expected outputs for the GPU test)
    T, Pn = 7, 24
    ang = torch.linspace(0, 2 * np.pi, Pn + 1)[:-1]
    yy, xx = torch.meshgrid(torch.arange(A, dtype=torch.float32), torch.arange(Bb, dtype=torch.float32), indexing="ij")
    cy, cx = 19.5, 27.0
    flow = torch.zeros(T, 2, A, Bb)
    for t in range(1, T):
        s = -0.04 * np.sin(np.pi * t / (T - 1)) * 3          # radial contraction, strongest mid-cycle
        flow[t, 0] = s * (xx - cx) + 0.2 * torch.randn(A, Bb, generator=g)      # channel 0: displacement along the last axis
        flow[t, 1] = s * (yy - cy) + 0.2 * torch.randn(A, Bb, generator=g)
    flow[0] = float("nan")
    contours = torch.zeros(2, T, Pn, 2)
    for si, r in enumerate((8.0, 12.0)):
        for t in range(T):
            k = 1 + (0 if t == 0 else -0.04 * np.sin(np.pi * t / (T - 1)) * 3)
            contours[si, t, :, 0] = cx + r * k * torch.cos(ang)
            contours[si, t, :, 1] = cy + r * k * torch.sin(ang)
    res = OS.from_ed(flow, contours, (1.25, 1.25), to_roll=2)
    con_all = torch.cat([contours[0], contours[1], contours[1][:, :10] + 1.0], dim=1)       # T, P_all, 2 (endo | epi | "rv")
    split = np.cumsum([Pn, Pn])
    out.update(ring_flow=flow.numpy(), ring_contours=contours.numpy(), ring_radial=res["radial_strain"].numpy(), ring_circ=res["circ_strain"].numpy(),
               ring_smooth=np.array(res["smooth"]), ring_con_all=con_all.numpy(), ring_split=split)
    for mode in ("from_ed_accumulation", "to_ed_accumulation", "to_ed"):
        out["ring_err_" + mode] = OS.contour_tracking_error(flow, con_all, split, mode)
    np.savez_compressed(os.path.join(HERE, "strain.npz"), **out)
    with open(os.path.join(HERE, "PIN_REPORT_strain.txt"), "w") as f:
        f.write("oracle/strain.py vs the reference (tests/golden/make_golden_strain.py)\n")
        for name, d, tol in REPORT:
            f.write("%-48s max|diff| = %.3e  (tol %.1e)\n" % (name, d, tol))
    print("wrote strain.npz (%d arrays), %d pins" % (len(out), len(REPORT)))


if __name__ == "__main__":
    main()
