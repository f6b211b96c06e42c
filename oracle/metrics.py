"""Oracle restatement of the downstream metrics that consume the path's output layout.

Test infrastructure only -- see oracle/__init__.py.  All `file:line` citations are relative to /root/reference.

Pinned against the reference (tests/golden/make_golden_metrics.py): ConfusionMatrix and every ratio of
nnunet/evaluation/metrics.py, the NaN / empty / full rules of its surface-distance wrappers, and the per-structure Jacobian
statistics of nnunet/compute_jacobian.py.  NOT pinned: the bodies of medpy.metric.binary.{hd, hd95, asd, assd} (medpy 0.4.0, absent)
and kornia.filters.spatial_gradient3d (kornia, absent) -- restated below from their published algorithms on scipy / numpy and
injected into the reference modules when the pins are generated.
"""
import numpy as np


# --------------------------------------------------------------------------- third-party restatements (parity unpinned)
class medpy_binary:
    """medpy.metric.binary (medpy 0.4.0): surface distances through scipy's exact Euclidean distance transform."""

    @staticmethod
    def surface_distances(result, reference, voxelspacing=None, connectivity=1):
        from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
        result = np.atleast_1d(np.asarray(result).astype(bool))
        reference = np.atleast_1d(np.asarray(reference).astype(bool))
        if voxelspacing is not None:
            voxelspacing = np.asarray(voxelspacing, dtype=np.float64)
            if voxelspacing.ndim == 0:
                voxelspacing = np.repeat(voxelspacing, result.ndim)
        footprint = generate_binary_structure(result.ndim, connectivity)
        if 0 == np.count_nonzero(result):
            raise RuntimeError("The first supplied array does not contain any binary object.")
        if 0 == np.count_nonzero(reference):
            raise RuntimeError("The second supplied array does not contain any binary object.")
        result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
        reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
        dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
        return dt[result_border]

    @classmethod
    def hd(cls, result, reference, voxelspacing=None, connectivity=1):
        return max(cls.surface_distances(result, reference, voxelspacing, connectivity).max(),
                   cls.surface_distances(reference, result, voxelspacing, connectivity).max())

    @classmethod
    def hd95(cls, result, reference, voxelspacing=None, connectivity=1):
        return np.percentile(np.hstack((cls.surface_distances(result, reference, voxelspacing, connectivity),
                                        cls.surface_distances(reference, result, voxelspacing, connectivity))), 95)

    @classmethod
    def asd(cls, result, reference, voxelspacing=None, connectivity=1):
        return cls.surface_distances(result, reference, voxelspacing, connectivity).mean()

    @classmethod
    def assd(cls, result, reference, voxelspacing=None, connectivity=1):
        return np.mean((cls.asd(result, reference, voxelspacing, connectivity), cls.asd(reference, result, voxelspacing, connectivity)))


def spatial_gradient3d(x):
    """kornia.filters.spatial_gradient3d(input, mode='diff', order=1): x [B, C, D, H, W] -> [B, C, 3, D, H, W]; replicate padding,
    0.5 * (x[+1] - x[-1]); component 0 along W, 1 along H, 2 along D."""
    x = np.asarray(x)
    p = np.pad(x, [(0, 0), (0, 0), (1, 1), (1, 1), (1, 1)], mode="edge")
    c = slice(1, -1)
    out = np.empty(x.shape[:2] + (3,) + x.shape[2:], dtype=x.dtype)
    out[:, :, 0] = p[:, :, c, c, 2:] - p[:, :, c, c, :-2]
    out[:, :, 1] = p[:, :, c, 2:, c] - p[:, :, c, :-2, c]
    out[:, :, 2] = p[:, :, 2:, c, c] - p[:, :, :-2, c, c]
    return 0.5 * out


# --------------------------------------------------------------------------- nnunet/evaluation/metrics.py
class ConfusionMatrix:
    """metrics.py:27-105."""

    def __init__(self, test=None, reference=None):
        self.test, self.reference = test, reference

    def get_matrix(self):
        t, r = self.test != 0, self.reference != 0
        assert t.shape == r.shape
        return int((t * r).sum()), int((t * ~r).sum()), int((~t * ~r).sum()), int((~t * r).sum())

    def get_existence(self):
        return (not np.any(self.test)), bool(np.all(self.test)), (not np.any(self.reference)), bool(np.all(self.reference))


def _ratio(test, reference, num, den, nan_when, nan_for_nonexisting=True):
    cm = ConfusionMatrix(test, reference)
    tp, fp, tn, fn = cm.get_matrix()
    te, tf, re_, rf = cm.get_existence()
    if nan_when(te, tf, re_, rf):
        return float("NaN") if nan_for_nonexisting else 0.0
    return float(num(tp, fp, tn, fn) / den(tp, fp, tn, fn))


def dice(test, reference, nan_for_nonexisting=True):
    """metrics.py:107-129."""
    return _ratio(test, reference, lambda tp, fp, tn, fn: 2.0 * tp, lambda tp, fp, tn, fn: 2 * tp + fp + fn, lambda te, tf, re_, rf: te and re_,
                  nan_for_nonexisting)


def jaccard(test, reference, nan_for_nonexisting=True):
    """metrics.py:132-147."""
    return _ratio(test, reference, lambda tp, fp, tn, fn: tp, lambda tp, fp, tn, fn: tp + fp + fn, lambda te, tf, re_, rf: te and re_, nan_for_nonexisting)


def precision(test, reference, nan_for_nonexisting=True):
    """metrics.py:150-165."""
    return _ratio(test, reference, lambda tp, fp, tn, fn: tp, lambda tp, fp, tn, fn: tp + fp, lambda te, tf, re_, rf: te, nan_for_nonexisting)


def sensitivity(test, reference, nan_for_nonexisting=True):
    """metrics.py:168-183."""
    return _ratio(test, reference, lambda tp, fp, tn, fn: tp, lambda tp, fp, tn, fn: tp + fn, lambda te, tf, re_, rf: re_, nan_for_nonexisting)


def specificity(test, reference, nan_for_nonexisting=True):
    """metrics.py:192-207."""
    return _ratio(test, reference, lambda tp, fp, tn, fn: tn, lambda tp, fp, tn, fn: tn + fp, lambda te, tf, re_, rf: rf, nan_for_nonexisting)


def accuracy(test, reference):
    """metrics.py:210-218."""
    tp, fp, tn, fn = ConfusionMatrix(test, reference).get_matrix()
    return float((tp + tn) / (tp + fp + tn + fn))


def _surface(fn, test, reference, voxel_spacing, connectivity, nan_for_nonexisting):
    te, tf, re_, rf = ConfusionMatrix(test, reference).get_existence()
    if te or tf or re_ or rf:
        return float("NaN") if nan_for_nonexisting else 0
    return fn(test, reference, voxel_spacing, connectivity)


def hausdorff_distance(test, reference, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1):
    """metrics.py:323-338."""
    return _surface(medpy_binary.hd, test, reference, voxel_spacing, connectivity, nan_for_nonexisting)


def hausdorff_distance_95(test, reference, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1):
    """metrics.py:341-356."""
    return _surface(medpy_binary.hd95, test, reference, voxel_spacing, connectivity, nan_for_nonexisting)


def avg_surface_distance(test, reference, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1):
    """metrics.py:359-374."""
    return _surface(medpy_binary.asd, test, reference, voxel_spacing, connectivity, nan_for_nonexisting)


def avg_surface_distance_symmetric(test, reference, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1):
    """metrics.py:377-392."""
    return _surface(medpy_binary.assd, test, reference, voxel_spacing, connectivity, nan_for_nonexisting)


# --------------------------------------------------------------------------- nnunet/compute_jacobian.py
def jacobian_determinant(disp):
    """compute_jacobian.py:16-60 for a 2-D field [H, W, 2] (np.gradient of displacement + identity grid)."""
    H, W = disp.shape[:2]
    grid = np.stack(np.meshgrid(np.arange(H), np.arange(W), indexing="ij"), 2)
    dfdx, dfdy = np.gradient(disp + grid)[:2]
    return dfdx[..., 0] * dfdy[..., 1] - dfdy[..., 0] * dfdx[..., 1]


def jacobian_frame_stats(frame_flow, frame_gt, names=("RV", "MYO", "LV")):
    """compute_jacobian.py:151-186: the per-frame row of the Jacobian table (structures are labels 1..len(names))."""
    jac = jacobian_determinant(frame_flow)
    res = {}
    for i, k in enumerate(names, 1):
        cur = jac[frame_gt == i]
        total, neg = float(cur.size), float((cur < 0).sum())
        res["abs(Mean jacobian - 1)_" + k] = abs(cur.mean() - 1)
        res["total_" + k] = total
        res["negative_" + k] = neg
        res["negative_%_" + k] = (neg / total) * 100
    res["abs(Mean jacobian - 1)_average"] = sum(res["abs(Mean jacobian - 1)_" + k] for k in names) / 3
    res["negative_%_average"] = sum(res["negative_%_" + k] for k in names) / 3
    res["abs(Mean jacobian - 1)"] = abs(jac.mean() - 1)
    res["total"] = float(jac.size)
    res["negative"] = float((jac < 0).sum())
    res["negative_%"] = (res["negative"] / res["total"]) * 100
    return res


def gradient_means(slice_flow):
    """compute_jacobian.py:146-159: slice_flow [T, H, W, 2] -> (temporal[T], spatial[T]) means of |spatial_gradient3d|."""
    g = np.abs(spatial_gradient3d(slice_flow.transpose(3, 0, 1, 2)[None]).astype(np.float64))
    gxy, gz = g[:, :, :2], g[:, :, 2]
    T = slice_flow.shape[0]
    return np.array([gz[:, :, t].mean() for t in range(T)]), np.array([gxy[:, :, :, t].mean() for t in range(T)])


# --------------------------------------------------------------------------- compute_SSIM*.py (skimage absent: parity unpinned)
def structural_similarity(im1, im2, data_range, win_size=7, full=False, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity (scikit-image >= 0.19; uniform window, use_sample_covariance=True) restated on
    scipy.ndimage.uniform_filter, in float64."""
    from scipy.ndimage import uniform_filter
    im1, im2 = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    NP = win_size ** im1.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(im1, size=win_size), uniform_filter(im2, size=win_size)
    uxx, uyy, uxy = uniform_filter(im1 * im1, size=win_size), uniform_filter(im2 * im2, size=win_size), uniform_filter(im1 * im2, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    mssim = S[pad:-pad, pad:-pad].mean(dtype=np.float64)
    return (mssim, S) if full else mssim
