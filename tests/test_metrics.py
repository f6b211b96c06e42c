"""Downstream metrics: the oracle against the reference's golden values on the CPU, the device path
(cineflow.metrics over the C ABI) against both on the GPU.

Tolerances: counts and ratios are integer arithmetic -> identical; surface distances are fp64 minima of the same expression ->
1e-12; Jacobian statistics inherit the fp32 determinant kernel -> 1e-5 relative; gradient means 1e-6."""
import numpy as np
import pytest

RATIOS = ("dice", "jaccard", "precision", "sensitivity", "specificity", "accuracy")
SURF = ("hausdorff_distance", "hausdorff_distance_95", "avg_surface_distance", "avg_surface_distance_symmetric")


def same(a, b, tol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), (a, b)
    d = np.abs(np.where(nan, 0.0, a - b)).max() if a.size else 0.0
    assert d <= tol, "max|diff| %.3e > %.1e" % (d, tol)


def nan_cases():
    e, f = np.zeros((6, 7), bool), np.ones((6, 7), bool)
    h = e.copy()
    h[2:4, 1:5] = True
    return {"empty_empty": (e, e), "empty_ref": (h, e), "empty_test": (e, h), "full_test": (f, h), "full_ref": (h, f)}


@pytest.mark.parametrize("tag", ["3d", "2d"])
def test_oracle_metrics_golden(golden, tag):
    from oracle import metrics as OM
    g = golden("metrics")
    test, gt, sp = g[tag + "_test"], g[tag + "_gt"], tuple(g[tag + "_spacing"])
    for c in (1, 2, 3):
        a, b = test == c, gt == c
        same([getattr(OM, n)(a, b) for n in RATIOS], g["%s_c%d_ratios" % (tag, c)])
        same([getattr(OM, n)(a, b, voxel_spacing=sp) for n in SURF], g["%s_c%d_surface" % (tag, c)])
    for k, (a, b) in nan_cases().items():
        same([getattr(OM, n)(a, b) for n in RATIOS + SURF], g["nan_" + k])


def test_oracle_jacobian_golden(golden):
    from oracle import metrics as OM
    g = golden("metrics")
    same(np.stack([OM.jacobian_determinant(f) for f in g["flow"]]), g["jac"])
    st = OM.jacobian_frame_stats(g["flow"][2], g["gt2"])
    same([st[k] for k in g["stats_keys"]], g["stats"], 1e-12)
    tg, sg = OM.gradient_means(g["flow"])
    same(tg, g["temporal_gradient"]), same(sg, g["spatial_gradient"])
    # spatial_gradient3d restatement: interior = central difference / 2, borders one-sided / 2 (replicate padding)
    x = np.random.default_rng(0).normal(size=(1, 1, 4, 5, 6)).astype(np.float32)
    gr = OM.spatial_gradient3d(x)
    assert np.allclose(gr[0, 0, 0, :, :, 2], 0.5 * (x[0, 0, :, :, 3] - x[0, 0, :, :, 1]))
    assert np.allclose(gr[0, 0, 2, 0], 0.5 * (x[0, 0, 1] - x[0, 0, 0]))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["3d", "2d"])
def test_metrics_device(dev, golden, tag):
    import torch
    from cineflow import metrics as M
    g = golden("metrics")
    test, gt, sp = g[tag + "_test"], g[tag + "_gt"], tuple(g[tag + "_spacing"])
    for c in (1, 2, 3):
        a, b = test == c, gt == c
        same([getattr(M, n)(a, b) for n in RATIOS], g["%s_c%d_ratios" % (tag, c)])
        same([getattr(M, n)(a, b, voxel_spacing=sp) for n in SURF], g["%s_c%d_surface" % (tag, c)], 1e-12)
        cm = M.ConfusionMatrix(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))      # device tensors, shared matrix
        assert M.dice(confusion_matrix=cm) == g["%s_c%d_ratios" % (tag, c)][0] and cm.get_size() == a.size
        assert M.recall(a, b) == M.sensitivity(a, b) and abs(M.false_positive_rate(a, b) - (1 - M.specificity(a, b))) == 0
        assert M.total_positives_test(a, b) == int(a.sum()) and M.total_negatives_reference(a, b) == int((~b).sum())
    # all classes at once
    K = 4
    h = M.label_confusion(test, gt, K)
    ref = np.array([[int(((test == t) & (gt == r)).sum()) for r in range(K)] for t in range(K)])
    assert np.array_equal(h, ref)
    for c in (1, 2, 3):
        assert 2.0 * h[c, c] / (h[c].sum() + h[:, c].sum()) == g["%s_c%d_ratios" % (tag, c)][0]
    with pytest.raises(ValueError):
        M.label_confusion(test, gt, 3)
    for k, (a, b) in nan_cases().items():
        same([getattr(M, n)(a, b) for n in RATIOS + SURF], g["nan_" + k])
    assert M.dice(np.zeros((3, 3)), np.zeros((3, 3)), nan_for_nonexisting=False) == 0.0
    with pytest.raises(AssertionError):
        M.dice(np.zeros((3, 3)), np.zeros((3, 4)))


@pytest.mark.gpu
def test_surface_distances_random_blobs(dev):
    """unequal spacings, objects touching the array border, single-voxel objects: against the scipy oracle"""
    from cineflow import metrics as M
    from oracle import metrics as OM
    rng = np.random.default_rng(11)
    for shape, sp in (((5, 30, 28), (7.5, 0.9, 1.3)), ((33, 31), (2.0, 0.5)), ((1, 20, 20), (3.0, 1.0, 1.0))):
        from scipy.ndimage import gaussian_filter
        a = gaussian_filter(rng.normal(size=shape), 2.0) > 0.02
        b = gaussian_filter(rng.normal(size=shape), 2.0) > 0.02
        a.flat[0] = True                      # a voxel in the corner
        for n in SURF:
            same(getattr(M, n)(a, b, voxel_spacing=sp), getattr(OM, n)(a, b, voxel_spacing=sp), 1e-12)
        d = M.surface_distances(a, b, sp).cpu().numpy()
        same(np.sort(d), np.sort(OM.medpy_binary.surface_distances(a, b, sp)), 1e-12)
    one = np.zeros((9, 9), bool)
    one[4, 4] = True
    two = np.zeros((9, 9), bool)
    two[1, 7] = True
    assert abs(M.hausdorff_distance(one, two) - np.hypot(3, 3)) < 1e-12
    with pytest.raises(NotImplementedError):
        M.hausdorff_distance(one, two, connectivity=2)


@pytest.mark.gpu
def test_jacobian_statistics_device(dev, golden):
    from cineflow import metrics as M
    g = golden("metrics")
    jac = np.stack([M.jacobian_determinant(f) for f in g["flow"]])
    assert jac.dtype == np.float64
    assert float(np.abs(jac - g["jac"]).max()) <= 2e-5
    st = M.jacobian_frame_stats(g["flow"][2], g["gt2"])
    keys = [str(k) for k in g["stats_keys"]]
    assert sorted(st) == sorted(keys)
    for k, v in zip(keys, g["stats"]):
        if k.startswith("total") or k.startswith("negative") and "%" not in k:
            assert abs(st[k] - v) <= 2, k                       # a determinant within 2e-5 of zero may change side
        elif "%" in k:
            assert abs(st[k] - v) <= 0.5, k
        else:
            assert abs(st[k] - v) <= 1e-4, k
    tg, sg = M.gradient_means(g["flow"])
    same(tg, g["temporal_gradient"], 1e-6), same(sg, g["spatial_gradient"], 1e-6)
    from oracle import metrics as OM
    x = np.random.default_rng(1).normal(size=(2, 2, 3, 9, 8)).astype(np.float32)
    same(M.spatial_gradient3d(x), OM.spatial_gradient3d(x), 1e-7)


def test_oracle_ssim_properties():
    """the restated SSIM (skimage absent): identical images score 1, the score is symmetric and falls with noise"""
    from oracle import metrics as OM
    rng = np.random.default_rng(3)
    a = rng.normal(size=(40, 36)) * 50 + 200
    assert abs(OM.structural_similarity(a, a, data_range=np.ptp(a)) - 1.0) < 1e-12
    b, c = a + rng.normal(size=a.shape) * 5, a + rng.normal(size=a.shape) * 25
    sb, sc = OM.structural_similarity(a, b, data_range=np.ptp(b)), OM.structural_similarity(a, c, data_range=np.ptp(b))
    assert 1.0 > sb > sc > 0.0 and abs(OM.structural_similarity(b, a, data_range=np.ptp(b)) - sb) < 1e-12


@pytest.mark.gpu
def test_ssim_device(dev):
    """device SSIM (compute_SSIM*.py) against the scipy restatement: score and full map, odd sizes, a non-default window"""
    from cineflow import metrics as M
    from oracle import metrics as OM
    rng = np.random.default_rng(4)
    for shape, win in (((64, 48), 7), ((33, 41), 7), ((20, 19), 11)):
        a = rng.normal(size=shape) * 40 + 300
        b = a + rng.normal(size=shape) * 12
        dr = b.max() - b.min()
        want, wmap = OM.structural_similarity(a, b, data_range=dr, win_size=win, full=True)
        got, gmap = M.structural_similarity(a, b, data_range=dr, win_size=win, full=True)
        assert abs(got - want) <= 1e-10 and float(np.abs(gmap - wmap).max()) <= 1e-9
        assert abs(M.structural_similarity(a.astype(np.float32), b.astype(np.float32), data_range=dr, win_size=win) - want) <= 1e-5
    assert abs(M.structural_similarity(a, a, data_range=1.0, win_size=11) - 1.0) <= 1e-12
    with pytest.raises(ValueError):
        M.structural_similarity(a, b)
    with pytest.raises(NotImplementedError):
        M.structural_similarity(np.zeros((3, 9, 9)), np.zeros((3, 9, 9)), data_range=1.0)
