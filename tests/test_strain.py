"""Contour tracking / strain (nnunet/get_strain.py): oracle against the reference's own output
(tests/golden/strain.npz, make_golden_strain.py) on the CPU, device path against both on the GPU."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "strain.npz")


def test_oracle_contour_sampler_matches_reference_output():
    from oracle import strain as OS
    g = np.load(GOLD)
    out = OS.spatial_transformer_contour(torch.from_numpy(g["stc_pts"]), torch.from_numpy(g["stc_field"]), g["stc_field"].shape[-2:])
    assert np.array_equal(out.numpy(), g["stc_out"])          # SpatialTransformerContour of the reference, bit for bit


def test_curvature_and_strain_known_answers():
    """a circle of radius r has curvature 1 / r everywhere; a ring scaled by k has Green-Lagrange strain (k^2 - 1) / 2 in both
    directions (parity unpinned for these two functions: get_strain.py itself cannot be imported here)."""
    from oracle import strain as OS
    th = np.linspace(0, 2 * np.pi, 400)
    c = OS.curvature(7.0 * np.cos(th), 7.0 * np.sin(th))
    assert np.allclose(c[5:-5], 1 / 7.0, rtol=1e-3)
    assert OS.smoothness_measure(np.arange(10.0), 3 * np.arange(10.0) + 1) == 0.0            # a straight line: zero curvature everywhere
    Pn = 32
    ang = torch.linspace(0, 2 * np.pi, Pn + 1)[:-1]
    pts = torch.zeros(3, 2, 2, Pn)
    for t, k in enumerate((1.0, 0.9, 1.2)):
        for s, r in enumerate((8.0, 12.0)):
            pts[t, s, 0], pts[t, s, 1] = 30 + r * k * torch.cos(ang), 25 + r * k * torch.sin(ang)
    radial, circ = OS.strain_curves(pts)
    want = torch.tensor([0.0, (0.81 - 1) / 2, (1.44 - 1) / 2])
    assert torch.allclose(radial, want, atol=1e-5) and torch.allclose(circ, want, atol=1e-5)
    r2, _ = OS.strain_curves(pts, to_roll=1)
    assert torch.allclose(r2, torch.roll(want, -1), atol=1e-5)


def test_oracle_reproduces_stored_ring():
    from oracle import strain as OS
    g = np.load(GOLD)
    res = OS.from_ed(g["ring_flow"], g["ring_contours"], (1.25, 1.25), to_roll=2)
    assert np.allclose(res["radial_strain"].numpy(), g["ring_radial"], atol=1e-6) and np.allclose(res["circ_strain"].numpy(), g["ring_circ"], atol=1e-6)
    assert abs(float(g["ring_radial"][np.argmin(g["ring_radial"])])) > 0.05           # the ring really contracts


@pytest.mark.gpu
def test_strain_device_vs_reference_and_oracle(dev):
    from cineflow import ops, strain as S
    from oracle import strain as OS
    g = np.load(GOLD)
    # the sampler against the reference's own output
    B, _, _, P = g["stc_pts"].shape
    out = ops.sample_points(torch.from_numpy(g["stc_field"]).to(dev), torch.from_numpy(g["stc_pts"]).reshape(B, 2, P).contiguous().to(dev)).cpu().numpy()
    assert float(np.abs(out - g["stc_out"][:, :, 0]).max()) <= 2e-6
    st = S.SpatialTransformerContour(size=g["stc_field"].shape[-2:])
    assert float(np.abs(st(torch.from_numpy(g["stc_pts"]), torch.from_numpy(g["stc_field"])).numpy() - g["stc_out"]).max()) <= 2e-6
    # strain curves and smoothness
    res = S.from_ed(g["ring_flow"], g["ring_contours"], (1.25, 1.25), to_roll=2)
    assert float(np.abs(res["radial_strain"].numpy() - g["ring_radial"]).max()) <= 1e-5
    assert float(np.abs(res["circ_strain"].numpy() - g["ring_circ"]).max()) <= 1e-5
    assert abs(res["smooth"] - float(g["ring_smooth"])) <= 1e-4 * max(1.0, abs(float(g["ring_smooth"])))
    # the three tracking-error modes
    for mode in ("from_ed_accumulation", "to_ed_accumulation", "to_ed"):
        e = S.contour_tracking_error(g["ring_flow"], g["ring_con_all"], g["ring_split"], mode)
        assert e.shape == (g["ring_flow"].shape[0] - 1, 3)
        assert float(np.abs(e - g["ring_err_" + mode]).max()) <= 1e-4
    # prepare_flow: the file layout of Postprocessed/Flow ([H,W,D,2] per frame) -> [D,T,2,W,H], NaN at ED, ED first
    frames = [np.random.default_rng(t).normal(size=(6, 5, 2, 2)).astype(np.float32) for t in range(4)]
    flow, idx = S.prepare_flow(frames, 2)
    assert flow.shape == (2, 5, 2, 5, 6) and idx.tolist() == [2, 3, 4, 0, 1] and np.isnan(flow[:, 0]).all()
    assert np.array_equal(flow[1, 1, 0], frames[2][:, :, 1, 0].T)
