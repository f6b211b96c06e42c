"""Downstream metrics on the device: the functions the reference's compute_metrics*.py and
compute_jacobian.py apply to the Segmentation / Registered / Flow outputs of the path.

Mirrors nnunet/evaluation/metrics.py (ConfusionMatrix and the metric functions, same names, arguments and NaN rules) and the
per-frame statistics of nnunet/compute_jacobian.py.  numpy arrays or device tensors in, Python floats out like the reference.
Counting, border extraction, nearest-border distances and reductions are HIP kernels (csrc/metrics.hip); the 95th percentile of
hausdorff_distance_95 is taken by numpy on the downloaded distance vector (a few thousand values).
"""
import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .ops import _f32, _stream, _u8


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _mask(a):
    """`a != 0` as a contiguous uint8 device tensor"""
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(_dev()) != 0).to(torch.uint8).contiguous()


def assert_shape(test, reference):
    assert test.shape == reference.shape, "Shape mismatch: {} and {}".format(test.shape, reference.shape)


class ConfusionMatrix:
    """metrics.py:27-105."""

    def __init__(self, test=None, reference=None):
        self.tp = self.fp = self.tn = self.fn = self.size = None
        self.reference_empty = self.reference_full = self.test_empty = self.test_full = None
        self.set_reference(reference)
        self.set_test(test)

    def set_test(self, test):
        self.test = test
        self.reset()

    def set_reference(self, reference):
        self.reference = reference
        self.reset()

    def reset(self):
        self.tp = self.fp = self.tn = self.fn = self.size = None
        self.test_empty = self.test_full = self.reference_empty = self.reference_full = None

    def compute(self):
        if self.test is None or self.reference is None:
            raise ValueError("'test' and 'reference' must both be set to compute confusion matrix.")
        assert_shape(self.test, self.reference)
        t, r = _mask(self.test), _mask(self.reference)
        n = t.numel()
        out = torch.empty(3, dtype=torch.int64, device=t.device)
        check(lib().cf_confusion_counts(_u8(t), _u8(r), n, out.data_ptr(), _stream()), "cf_confusion_counts")
        self.tp, self.fp, self.fn = (int(v) for v in out.cpu().tolist())
        self.tn = n - self.tp - self.fp - self.fn
        self.size = n
        self.test_empty, self.test_full = self.tp + self.fp == 0, self.tp + self.fp == n
        self.reference_empty, self.reference_full = self.tp + self.fn == 0, self.tp + self.fn == n

    def get_matrix(self):
        if None in (self.tp, self.fp, self.tn, self.fn):
            self.compute()
        return self.tp, self.fp, self.tn, self.fn

    def get_size(self):
        if self.size is None:
            self.compute()
        return self.size

    def get_existence(self):
        if None in (self.test_empty, self.test_full, self.reference_empty, self.reference_full):
            self.compute()
        return self.test_empty, self.test_full, self.reference_empty, self.reference_full


def _nan(nan_for_nonexisting):
    return float("NaN") if nan_for_nonexisting else 0.


def _cm(test, reference, confusion_matrix):
    return ConfusionMatrix(test, reference) if confusion_matrix is None else confusion_matrix


def dice(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """2TP / (2TP + FP + FN)  (metrics.py:107-129)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    test_empty, test_full, reference_empty, reference_full = cm.get_existence()
    if test_empty and reference_empty:
        return _nan(nan_for_nonexisting)
    return float(2. * tp / (2 * tp + fp + fn))


def jaccard(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TP / (TP + FP + FN)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    test_empty, test_full, reference_empty, reference_full = cm.get_existence()
    if test_empty and reference_empty:
        return _nan(nan_for_nonexisting)
    return float(tp / (tp + fp + fn))


def precision(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TP / (TP + FP)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    if cm.get_existence()[0]:
        return _nan(nan_for_nonexisting)
    return float(tp / (tp + fp))


def sensitivity(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TP / (TP + FN)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    if cm.get_existence()[2]:
        return _nan(nan_for_nonexisting)
    return float(tp / (tp + fn))


def recall(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TP / (TP + FN)"""
    return sensitivity(test, reference, confusion_matrix, nan_for_nonexisting, **kwargs)


def specificity(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TN / (TN + FP)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    if cm.get_existence()[3]:
        return _nan(nan_for_nonexisting)
    return float(tn / (tn + fp))


def accuracy(test=None, reference=None, confusion_matrix=None, **kwargs):
    """(TP + TN) / (TP + FP + FN + TN)"""
    tp, fp, tn, fn = _cm(test, reference, confusion_matrix).get_matrix()
    return float((tp + tn) / (tp + fp + tn + fn))


def fscore(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, beta=1., **kwargs):
    """(1 + b^2) * TP / ((1 + b^2) * TP + b^2 * FN + FP)"""
    precision_ = precision(test, reference, confusion_matrix, nan_for_nonexisting)
    recall_ = recall(test, reference, confusion_matrix, nan_for_nonexisting)
    return (1 + beta * beta) * precision_ * recall_ / ((beta * beta * precision_) + recall_)


def false_positive_rate(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """FP / (FP + TN)"""
    return 1 - specificity(test, reference, confusion_matrix, nan_for_nonexisting)


def false_omission_rate(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """FN / (TN + FN)"""
    cm = _cm(test, reference, confusion_matrix)
    tp, fp, tn, fn = cm.get_matrix()
    if cm.get_existence()[1]:
        return _nan(nan_for_nonexisting)
    return float(fn / (fn + tn))


def false_negative_rate(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """FN / (TP + FN)"""
    return 1 - sensitivity(test, reference, confusion_matrix, nan_for_nonexisting)


def true_negative_rate(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TN / (TN + FP)"""
    return specificity(test, reference, confusion_matrix, nan_for_nonexisting)


def false_discovery_rate(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """FP / (TP + FP)"""
    return 1 - precision(test, reference, confusion_matrix, nan_for_nonexisting)


def negative_predictive_value(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, **kwargs):
    """TN / (TN + FN)"""
    return 1 - false_omission_rate(test, reference, confusion_matrix, nan_for_nonexisting)


def total_positives_test(test=None, reference=None, confusion_matrix=None, **kwargs):
    """TP + FP"""
    tp, fp, tn, fn = _cm(test, reference, confusion_matrix).get_matrix()
    return tp + fp


def total_negatives_test(test=None, reference=None, confusion_matrix=None, **kwargs):
    """TN + FN"""
    tp, fp, tn, fn = _cm(test, reference, confusion_matrix).get_matrix()
    return tn + fn


def total_positives_reference(test=None, reference=None, confusion_matrix=None, **kwargs):
    """TP + FN"""
    tp, fp, tn, fn = _cm(test, reference, confusion_matrix).get_matrix()
    return tp + fn


def total_negatives_reference(test=None, reference=None, confusion_matrix=None, **kwargs):
    """TN + FP"""
    tp, fp, tn, fn = _cm(test, reference, confusion_matrix).get_matrix()
    return tn + fp


# ------------------------------------------------------------------------------------------------ surface distances
def _border(mask):
    shape = tuple(mask.shape)
    assert len(shape) in (2, 3), "surface distances are built for 2-D and 3-D masks"
    D, H, W = (1,) * (3 - len(shape)) + shape
    coords = torch.empty(3 * mask.numel(), dtype=torch.int32, device=mask.device)
    count = torch.empty(1, dtype=torch.int32, device=mask.device)
    check(lib().cf_surface_border(_u8(mask), D, H, W, len(shape), coords.data_ptr(), count.data_ptr(), _stream()), "cf_surface_border")
    return coords, int(count.item())


def surface_distances(result, reference, voxelspacing=None, connectivity=1):
    """medpy.metric.binary.__surface_distances: distance of every border voxel of `result` to the nearest border voxel of
    `reference` -> fp64 device tensor (in no particular order)."""
    if connectivity != 1:
        raise NotImplementedError("surface distances are built for connectivity 1 (the reference's only value)")
    a, b = _mask(result), _mask(reference)
    assert_shape(a, b)
    nd = a.dim()
    sp = np.ones(nd) if voxelspacing is None else np.asarray(voxelspacing, dtype=np.float64) * np.ones(nd)
    sz, sy, sx = ([1.0] * (3 - nd) + [float(v) for v in sp])
    ca, na = _border(a)
    cb, nb = _border(b)
    if na == 0:
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if nb == 0:
        raise RuntimeError("The second supplied array does not contain any binary object.")
    dist = torch.empty(na, dtype=torch.float64, device=a.device)
    check(lib().cf_surface_min_dist(ca.data_ptr(), na, cb.data_ptr(), nb, sz, sy, sx, dist.data_ptr(), _stream()), "cf_surface_min_dist")
    return dist


def _max_sum(dist):
    out = torch.empty(2, dtype=torch.float64, device=dist.device)
    check(lib().cf_max_sum_nonneg(dist.data_ptr(), dist.numel(), out.data_ptr(), _stream()), "cf_max_sum_nonneg")
    mx, sm = out.cpu().tolist()
    return mx, sm


def _surface_metric(kind, test, reference, confusion_matrix, nan_for_nonexisting, voxel_spacing, connectivity):
    cm = _cm(test, reference, confusion_matrix)
    test_empty, test_full, reference_empty, reference_full = cm.get_existence()
    if test_empty or test_full or reference_empty or reference_full:
        return float("NaN") if nan_for_nonexisting else 0
    test, reference = cm.test, cm.reference
    d1 = surface_distances(test, reference, voxel_spacing, connectivity)
    if kind == "asd":
        return _max_sum(d1)[1] / d1.numel()
    d2 = surface_distances(reference, test, voxel_spacing, connectivity)
    if kind == "hd":
        return max(_max_sum(d1)[0], _max_sum(d2)[0])
    if kind == "assd":
        return float(np.mean((_max_sum(d1)[1] / d1.numel(), _max_sum(d2)[1] / d2.numel())))
    return float(np.percentile(np.hstack((d1.cpu().numpy(), d2.cpu().numpy())), 95))


def hausdorff_distance(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1, **kwargs):
    """metrics.py:323-338 (medpy.metric.hd)."""
    return _surface_metric("hd", test, reference, confusion_matrix, nan_for_nonexisting, voxel_spacing, connectivity)


def hausdorff_distance_95(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1, **kwargs):
    """metrics.py:341-356 (medpy.metric.hd95)."""
    return _surface_metric("hd95", test, reference, confusion_matrix, nan_for_nonexisting, voxel_spacing, connectivity)


def avg_surface_distance(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1, **kwargs):
    """metrics.py:359-374 (medpy.metric.asd)."""
    return _surface_metric("asd", test, reference, confusion_matrix, nan_for_nonexisting, voxel_spacing, connectivity)


def avg_surface_distance_symmetric(test=None, reference=None, confusion_matrix=None, nan_for_nonexisting=True, voxel_spacing=None, connectivity=1,
                                   **kwargs):
    """metrics.py:377-392 (medpy.metric.assd)."""
    return _surface_metric("assd", test, reference, confusion_matrix, nan_for_nonexisting, voxel_spacing, connectivity)


ALL_METRICS = {
    "False Positive Rate": false_positive_rate, "Dice": dice, "Jaccard": jaccard, "Hausdorff Distance": hausdorff_distance,
    "Hausdorff Distance 95": hausdorff_distance_95, "Precision": precision, "Recall": recall,
    "Avg. Symmetric Surface Distance": avg_surface_distance_symmetric, "Avg. Surface Distance": avg_surface_distance, "Accuracy": accuracy,
    "False Omission Rate": false_omission_rate, "Negative Predictive Value": negative_predictive_value,
    "False Negative Rate": false_negative_rate, "True Negative Rate": true_negative_rate, "False Discovery Rate": false_discovery_rate,
    "Total Positives Test": total_positives_test, "Total Negatives Test": total_negatives_test,
    "Total Positives Reference": total_positives_reference, "total Negatives Reference": total_negatives_reference,
}


def label_confusion(test, reference, num_classes):
    """All classes of nnunet/compute_metrics.py:96-106 in one pass: K x K int64 matrix M[t, r] = #voxels with test label t and
    reference label r (labels must be < num_classes <= 16); per-class Dice = 2 M[c,c] / (M[c,:].sum() + M[:,c].sum())."""
    t = (test if torch.is_tensor(test) else torch.from_numpy(np.ascontiguousarray(test))).to(_dev()).to(torch.uint8).contiguous()
    r = (reference if torch.is_tensor(reference) else torch.from_numpy(np.ascontiguousarray(reference))).to(_dev()).to(torch.uint8).contiguous()
    assert_shape(t, r)
    K = int(num_classes)
    hist = torch.empty(K * K + 1, dtype=torch.int64, device=t.device)
    check(lib().cf_label_confusion(_u8(t), _u8(r), t.numel(), K, hist.data_ptr(), _stream()), "cf_label_confusion")
    h = hist.cpu().numpy()
    if h[-1]:
        raise ValueError("label_confusion: %d voxels carry a label >= num_classes=%d" % (int(h[-1]), K))
    return h[:-1].reshape(K, K)


# ------------------------------------------------------------------------------------------------ compute_jacobian.py
def jacobian_determinant(disp):
    """compute_jacobian.py:16-60 for a 2-D displacement field [H, W, 2] (channel-last like the script) -> float64 [H, W] (numpy for a
    numpy field, device tensor for a tensor)."""
    was_numpy = not torch.is_tensor(disp)
    d = torch.from_numpy(np.ascontiguousarray(disp)) if was_numpy else disp
    assert d.dim() == 3 and d.shape[-1] == 2, "flow has to be [H, W, 2]"
    flow = d.to(_dev(), dtype=torch.float32).permute(2, 0, 1)[None].contiguous()
    jac = ops.jacobian_det(flow)[0]
    return jac.cpu().numpy() if was_numpy else jac


def jacobian_frame_stats(frame_flow, frame_gt, names=("RV", "MYO", "LV")):
    """compute_jacobian.py:151-186: the per-frame row of the Jacobian table (structures = labels 1..len(names) of frame_gt)."""
    jac = jacobian_determinant(frame_flow)
    jac_t = jac if torch.is_tensor(jac) else torch.from_numpy(np.ascontiguousarray(jac))
    jac_t = jac_t.to(_dev(), dtype=torch.float64).contiguous()
    gt = (frame_gt if torch.is_tensor(frame_gt) else torch.from_numpy(np.ascontiguousarray(frame_gt))).to(_dev())
    K = len(names) + 1
    lab = torch.where((gt >= 0) & (gt < K), gt, torch.full_like(gt, 255)).to(torch.uint8).contiguous()
    stats = torch.empty(3 * K, dtype=torch.float64, device=lab.device)
    check(lib().cf_region_stats(jac_t.data_ptr(), _u8(lab), lab.numel(), K, stats.data_ptr(), _stream()), "cf_region_stats")
    whole = torch.empty(3, dtype=torch.float64, device=lab.device)
    zeros = torch.zeros_like(lab)
    check(lib().cf_region_stats(jac_t.data_ptr(), _u8(zeros), lab.numel(), 1, whole.data_ptr(), _stream()), "cf_region_stats")
    st = stats.cpu().numpy().reshape(K, 3)
    res = {}
    for i, k in enumerate(names, 1):
        s, total, neg = st[i]
        res["abs(Mean jacobian - 1)_" + k] = abs(s / total - 1) if total else float("nan")
        res["total_" + k] = float(total)
        res["negative_" + k] = float(neg)
        res["negative_%_" + k] = (neg / total) * 100 if total else float("nan")
    res["abs(Mean jacobian - 1)_average"] = sum(res["abs(Mean jacobian - 1)_" + k] for k in names) / 3
    res["negative_%_average"] = sum(res["negative_%_" + k] for k in names) / 3
    s, total, neg = whole.cpu().tolist()
    res["abs(Mean jacobian - 1)"] = abs(s / total - 1)
    res["total"] = float(total)
    res["negative"] = float(neg)
    res["negative_%"] = (neg / total) * 100
    return res


def spatial_gradient3d(x):
    """kornia.filters.spatial_gradient3d(x, mode='diff', order=1): [B, C, D, H, W] fp32 -> [B, C, 3, D, H, W]."""
    was_numpy = not torch.is_tensor(x)
    t = (torch.from_numpy(np.ascontiguousarray(x)) if was_numpy else x).to(_dev(), dtype=torch.float32).contiguous()
    B, C, D, H, W = t.shape
    out = torch.empty((B, C, 3, D, H, W), dtype=torch.float32, device=t.device)
    check(lib().cf_spatial_gradient3d(_f32(t), _f32(out), B * C, D, H, W, _stream()), "cf_spatial_gradient3d")
    return out.cpu().numpy() if was_numpy else out


def gradient_means(slice_flow):
    """compute_jacobian.py:146-159: slice_flow [T, H, W, 2] -> (temporal gradient [T], spatial gradient [T]): the per-frame means
    of |d/dt| and of |d/dx|, |d/dy| of both flow components."""
    t = (slice_flow if torch.is_tensor(slice_flow) else torch.from_numpy(np.ascontiguousarray(slice_flow))).to(_dev(), dtype=torch.float32)
    T, H, W, _ = t.shape
    g = spatial_gradient3d(t.permute(3, 0, 1, 2)[None].contiguous())          # [1, 2, 3, T, H, W]
    sums = torch.empty(2 * 3 * T, dtype=torch.float64, device=g.device)         # slabs (c, comp) x frame
    check(lib().cf_slab_abs_sum(_f32(g), 6, 1, T, H * W, sums.data_ptr(), _stream()), "cf_slab_abs_sum")
    s = sums.cpu().numpy().reshape(2, 3, T)
    return s[:, 2].sum(0) / (2 * H * W), s[:, :2].sum((0, 1)) / (4 * H * W)


# ------------------------------------------------------------------------------------------------ compute_SSIM*.py
def structural_similarity(im1, im2, *, win_size=None, data_range=None, full=False, K1=0.01, K2=0.03, use_sample_covariance=True):
    """skimage.metrics.structural_similarity as nnunet/compute_SSIM.py:91 and compute_SSIM_crop_per_structure.py:88 call it: 2-D
    images, uniform win_size x win_size window (default 7), `data_range` given.  Computed in fp64 (skimage keeps float32 inputs in
    float32).  Returns the mean SSIM over the interior (the map cropped by (win_size - 1) // 2), and with full=True also the map."""
    a = (im1 if torch.is_tensor(im1) else torch.from_numpy(np.ascontiguousarray(im1))).to(_dev(), dtype=torch.float64).contiguous()
    b = (im2 if torch.is_tensor(im2) else torch.from_numpy(np.ascontiguousarray(im2))).to(_dev(), dtype=torch.float64).contiguous()
    assert_shape(a, b)
    if a.dim() != 2:
        raise NotImplementedError("structural_similarity is built for 2-D images (the reference scores slice by slice)")
    if data_range is None:
        raise ValueError("data_range must be given for floating-point images (as skimage requires)")
    win = 7 if win_size is None else int(win_size)
    H, W = a.shape
    if win % 2 == 0 or win > min(H, W):
        raise ValueError("win_size must be odd and no larger than the image")
    NP = win * win
    cov_norm = NP / (NP - 1.0) if use_sample_covariance else 1.0
    R = float(data_range)
    S = torch.empty((H, W), dtype=torch.float64, device=a.device)
    check(lib().cf_ssim_map(a.data_ptr(), b.data_ptr(), H, W, win, (K1 * R) ** 2, (K2 * R) ** 2, cov_norm, S.data_ptr(), _stream()), "cf_ssim_map")
    pad = (win - 1) // 2
    inner = S[pad:H - pad, pad:W - pad].contiguous()
    zeros = torch.zeros(inner.numel(), dtype=torch.uint8, device=a.device)
    st = torch.empty(3, dtype=torch.float64, device=a.device)
    check(lib().cf_region_stats(inner.data_ptr(), _u8(zeros), inner.numel(), 1, st.data_ptr(), _stream()), "cf_region_stats")
    s, cnt, _ = st.cpu().tolist()
    mssim = s / cnt
    if full:
        return mssim, (S if torch.is_tensor(im1) else S.cpu().numpy())
    return mssim
