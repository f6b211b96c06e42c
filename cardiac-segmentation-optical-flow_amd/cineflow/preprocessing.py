"""Test-time preprocessing on the device -- the caller side of the hot path.

Mirrors the reference interface of nnunet/preprocessing/cropping.py and nnunet/preprocessing/preprocessing.py (same function and
class names, argument meaning and return values; numpy in, numpy out like the reference, device tensors kept when given) so that
`trainer.preprocess_patient` (nnunet/training/network_training/nnUNetTrainer.py:571-597) is a drop-in.  Every number comes from
libcineflow_hip.so (csrc/preprocess.hip, csrc/postprocess.hip); torch only holds memory, slices and moves it.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .nifti import read_nifti
from .ops import _f32, _stream, _u8

RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3  # nnunet/configuration.py


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, dtype=torch.float32):
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(_dev(), dtype=dtype).contiguous()


# ------------------------------------------------------------------------------------------------ cropping.py
def create_nonzero_mask(data):
    """cropping.py:25-32: OR over the modalities of data != 0, then scipy's binary_fill_holes.  data (C, X, Y, Z) or (C, X, Y);
    returns a bool array of the same kind (numpy / device tensor) as `data`."""
    was_numpy = not torch.is_tensor(data)
    t = _to_dev(data)
    assert t.dim() in (3, 4), "data must have shape (C, X, Y, Z) or shape (C, X, Y)"
    C = t.shape[0]
    shape = tuple(t.shape[1:])
    D, H, W = (1,) * (3 - len(shape)) + shape
    n = D * H * W
    mask = torch.empty(n, dtype=torch.uint8, device=t.device)
    check(lib().cf_nonzero_mask(_f32(t), C, n, _u8(mask), _stream()), "cf_nonzero_mask")
    # background components by the connected-component sweeps of the export step, then the hole fill
    labels = torch.empty(n, dtype=torch.int32, device=t.device)
    changed = torch.zeros(1, dtype=torch.int32, device=t.device)
    zero = (ctypes.c_uint8 * 1)(0)
    check(lib().cf_cc_init(_u8(mask), labels.data_ptr(), n, ctypes.cast(zero, ctypes.c_void_p), 1, _stream()), "cf_cc_init")
    while True:
        changed.zero_()
        for _ in range(8):
            check(lib().cf_cc_sweep(labels.data_ptr(), D, H, W, changed.data_ptr(), _stream()), "cf_cc_sweep")
        if int(changed.item()) == 0:
            break
    touch = torch.empty(n, dtype=torch.int32, device=t.device)
    check(lib().cf_fill_holes(_u8(mask), labels.data_ptr(), touch.data_ptr(), D, H, W, len(shape), _stream()), "cf_fill_holes")
    out = mask.view(shape).bool()
    return out.cpu().numpy() if was_numpy else out


def get_bbox_from_mask(mask, outside_value=0):
    """cropping.py:47-55 -> [[minz, maxz+1], [minx, maxx+1], [miny, maxy+1]] as Python ints."""
    t = mask if torch.is_tensor(mask) else torch.from_numpy(np.ascontiguousarray(mask))
    t = (t != outside_value).to(_dev(), dtype=torch.uint8).contiguous()
    assert t.dim() == 3, "only supports 3d images"
    bbox = torch.empty(6, dtype=torch.int32, device=t.device)
    check(lib().cf_mask_bbox(_u8(t), t.shape[0], t.shape[1], t.shape[2], bbox.data_ptr(), _stream()), "cf_mask_bbox")
    b = bbox.cpu().tolist()
    if b[1] < 0:
        raise ValueError("get_bbox_from_mask: the mask is empty")   # np.min of an empty sequence raises in the reference
    return [[b[0], b[1] + 1], [b[2], b[3] + 1], [b[4], b[5] + 1]]


def crop_to_bbox(image, bbox):
    """cropping.py:58-61."""
    assert len(image.shape) == 3, "only supports 3d images"
    return image[bbox[0][0]:bbox[0][1], bbox[1][0]:bbox[1][1], bbox[2][0]:bbox[2][1]]


def crop_to_nonzero(data, seg=None, nonzero_label=-1):
    """cropping.py:104-137 -> (data, seg, bbox); seg is created (0 inside the mask, nonzero_label outside) when absent."""
    was_numpy = not torch.is_tensor(data)
    t = _to_dev(data)
    nonzero_mask = create_nonzero_mask(t)
    bbox = get_bbox_from_mask(nonzero_mask, 0)
    sl = (slice(None), slice(*bbox[0]), slice(*bbox[1]), slice(*bbox[2]))
    t = t[sl].contiguous()
    m = nonzero_mask[sl[1:]].to(torch.uint8).contiguous()
    # a created segmentation is the zero map with `nonzero_label` outside the mask (cropping.py:131-135)
    s = _to_dev(seg)[sl].contiguous() if seg is not None else torch.zeros((1,) + tuple(m.shape), dtype=torch.float32, device=t.device)
    check(lib().cf_seg_outside_mask(_f32(s), _u8(m), s.shape[0], m.numel(), float(nonzero_label), _stream()), "cf_seg_outside_mask")
    seg_dtype = seg.dtype if seg is not None else (np.int64 if was_numpy else torch.int64)
    if was_numpy:
        return t.cpu().numpy(), s.cpu().numpy().astype(seg_dtype), bbox
    return t, s.to(seg_dtype), bbox


def load_case_from_list_of_files(data_files, seg_file=None, info_dict=None):
    """cropping.py:74-101 with the package's NIfTI-1 reader in place of SimpleITK (absent from the image)."""
    assert isinstance(data_files, (list, tuple)), "case must be either a list or a tuple"
    properties = OrderedDict()
    arrays, first = [], None
    for f in data_files:
        a, pr = read_nifti(f)
        arrays.append(a[None])
        first = first or pr
    properties["original_size_of_raw_data"] = np.array(arrays[0].shape[1:])
    properties["original_spacing"] = np.array(first["itk_spacing"])[[2, 1, 0]]
    properties["list_of_data_files"] = data_files
    properties["seg_file"] = seg_file
    properties["volume_per_voxel"] = float(np.prod(first["itk_spacing"], dtype=np.float64))
    properties["itk_origin"] = first["itk_origin"]
    properties["itk_spacing"] = first["itk_spacing"]
    properties["itk_direction"] = first["itk_direction"]
    if info_dict is not None:
        properties.update(info_dict)
    data_npy = np.vstack(arrays)
    seg_npy = read_nifti(seg_file)[0][None].astype(np.float32) if seg_file is not None else None
    return data_npy.astype(np.float32), seg_npy, properties


class ImageCropper(object):
    """cropping.py:145-177 (the test-time entry points)."""

    @staticmethod
    def crop(data, properties, seg=None):
        data, seg, bbox = crop_to_nonzero(data, seg, nonzero_label=-1)
        properties["crop_bbox"] = bbox
        properties["classes"] = torch.unique(seg).cpu().numpy() if torch.is_tensor(seg) else np.unique(seg)
        seg[seg < -1] = 0
        properties["size_after_cropping"] = tuple(data[0].shape)
        return data, seg, properties

    @staticmethod
    def crop_from_list_of_files(data_files, seg_file=None, info_dict=None):
        data, seg, properties = load_case_from_list_of_files(data_files, seg_file, info_dict)
        return ImageCropper.crop(data, properties, seg)


# ------------------------------------------------------------------------------------------------ preprocessing.py
def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """preprocessing.py:30-32."""
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    """preprocessing.py:35-37."""
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def _spline_axis(x, axis, m):
    """one cubic-spline pass: x fp64 device tensor, `axis` resampled to m samples"""
    if x.shape[axis] == m:
        return x
    x = x.contiguous()
    n = x.shape[axis]
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
    out = torch.empty(x.shape[:axis] + (m,) + x.shape[axis + 1:], dtype=torch.float64, device=x.device)
    check(lib().cf_spline3_resample_axis(x.data_ptr(), out.data_ptr(), outer, n, inner, m, _stream()), "cf_spline3_resample_axis")
    return out


def _resize_cubic(t, new_shape, slab_axis):
    """skimage resize(order=3, mode='edge', clip=True, anti_aliasing=False) of every channel of t [C, x, y, z] (fp32 device):
    per 2-D slice along `slab_axis` (1..3, that axis keeps its size) or of the whole 3-D channel (slab_axis None)."""
    C = t.shape[0]
    x = t.double()
    dims = list(x.shape)
    if slab_axis is None:
        A, S, B = 1, 1, int(np.prod(dims[1:], dtype=np.int64))
    else:
        A, S, B = int(np.prod(dims[1:slab_axis], dtype=np.int64)), dims[slab_axis], int(np.prod(dims[slab_axis + 1:], dtype=np.int64))
    mm = torch.empty(C * S * 2, dtype=torch.float64, device=x.device)
    part = torch.empty(C * S * 2 * lib().cf_slab_minmax_chunks(C, A, S, B), dtype=torch.float64, device=x.device)
    check(lib().cf_slab_minmax(x.data_ptr(), C, A, S, B, mm.data_ptr(), part.data_ptr(), _stream()), "cf_slab_minmax")
    for a in (1, 2, 3):
        if a != slab_axis:
            x = _spline_axis(x, a, int(new_shape[a - 1]))
    dims = list(x.shape)
    if slab_axis is None:
        A, S, B = 1, 1, int(np.prod(dims[1:], dtype=np.int64))
    else:
        A, S, B = int(np.prod(dims[1:slab_axis], dtype=np.int64)), dims[slab_axis], int(np.prod(dims[slab_axis + 1:], dtype=np.int64))
    out = torch.empty(dims, dtype=torch.float32, device=x.device)
    check(lib().cf_slab_clip_to_f32(x.contiguous().data_ptr(), _f32(out), C, A, S, B, mm.data_ptr(), _stream()), "cf_slab_clip_to_f32")
    return out


def _resize_labels(t, new_shape, lin):
    """batchgenerators' resize_segmentation for order 1: every label's indicator is resized (linear on the axes flagged in `lin`,
    nearest on the others) and the label assigned where it reaches 0.5, labels ascending."""
    out = torch.zeros((t.shape[0],) + tuple(int(v) for v in new_shape), dtype=torch.float32, device=t.device)
    for c in torch.unique(t).tolist():
        ind = ops.resize3d((t == c).float().contiguous(), new_shape, lin)
        check(lib().cf_assign_where_ge(_f32(out), _f32(ind), out.numel(), 0.5, float(c), _stream()), "cf_assign_where_ge")
    return out


def resample_data_or_seg(data, new_shape, is_seg, axis=None, order=3, do_separate_z=False, order_z=0):
    """preprocessing.py:111-200.  data (c, x, y, z), numpy or device tensor (the same kind is returned).  Built: data of order
    0, 1 or 3 and segmentations of order 0 or 1, with order_z 0 or 1 (data) / 0 (segmentations) along the separate axis."""
    was_numpy = not torch.is_tensor(data)
    t = torch.from_numpy(np.ascontiguousarray(data)) if was_numpy else data
    dtype_in = t.dtype
    assert t.dim() == 4 and len(new_shape) == 3, "data must be (c, x, y, z)"
    new_shape = tuple(int(v) for v in new_shape)
    if tuple(t.shape[1:]) == new_shape:
        return data
    if order not in (0, 1, 3) or order_z not in (0, 1):
        raise NotImplementedError("resampling is built for interpolation orders 0, 1 and 3 in-plane and 0 / 1 along z (got %s / %s)" % (order, order_z))
    if is_seg and (order not in (0, 1) or (do_separate_z and order_z != 0)):
        raise NotImplementedError("segmentation resampling is built for orders 0 and 1 (order_z 0)")
    t = t.to(_dev(), dtype=torch.float32).contiguous()
    sep = None
    if do_separate_z:
        assert len(axis) == 1, "only one anisotropic axis supported"
        sep = int(axis[0])
    lin = [int(order != 0)] * 3
    if sep is not None:
        lin[sep] = int(order_z)
    if is_seg:
        out = ops.resize3d(t, new_shape, [0, 0, 0]) if order == 0 else _resize_labels(t, new_shape, lin)
    elif order == 3:
        if sep is None:
            out = _resize_cubic(t, new_shape, None)
        else:
            inplane = list(new_shape)
            inplane[sep] = t.shape[1 + sep]
            out = _resize_cubic(t, inplane, 1 + sep)            # every slice on its own, clipped to its own range, cast to fp32
            if t.shape[1 + sep] != new_shape[sep]:
                zl = [0, 0, 0]
                zl[sep] = int(order_z)
                out = ops.resize3d(out, new_shape, zl)          # the map_coordinates pass along the separate axis
    else:
        out = ops.resize3d(t, new_shape, lin)
    if is_seg and order == 0:
        out = out.round()
    out = out.to(dtype_in)
    return out.cpu().numpy() if was_numpy else out


def resample_patient(data, seg, original_spacing, target_spacing, order_data=3, order_seg=0, force_separate_z=False, order_z_data=0,
                     order_z_seg=0, separate_z_anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """preprocessing.py:40-108."""
    assert not ((data is None) and (seg is None))
    if data is not None:
        assert len(data.shape) == 4, "data must be c x y z"
    if seg is not None:
        assert len(seg.shape) == 4, "seg must be c x y z"
    shape = np.array(data[0].shape if data is not None else seg[0].shape)
    new_shape = np.round(((np.array(original_spacing) / np.array(target_spacing)).astype(float) * shape)).astype(int)
    if force_separate_z is not None:
        do_separate_z = force_separate_z
        axis = get_lowres_axis(original_spacing) if force_separate_z else None
    elif get_do_separate_z(original_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(original_spacing)
    elif get_do_separate_z(target_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(target_spacing)
    else:
        do_separate_z, axis = False, None
    if axis is not None and len(axis) != 1:
        do_separate_z = False       # (0.24, 1.25, 1.25)-like spacings: no separate out-of-plane pass
    data_reshaped = resample_data_or_seg(data, new_shape, False, axis, order_data, do_separate_z, order_z=order_z_data) if data is not None else None
    seg_reshaped = resample_data_or_seg(seg, new_shape, True, axis, order_seg, do_separate_z, order_z=order_z_seg) if seg is not None else None
    return data_reshaped, seg_reshaped


def _moments(x, seg, lo=None, hi=None):
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    use_range = lo is not None
    check(lib().cf_masked_moments(_f32(x), None if seg is None else _f32(seg), x.numel(), int(use_range), float(lo or 0.0), float(hi or 0.0),
                                  out.data_ptr(), _stream()), "cf_masked_moments")
    s1, s2, n = out.cpu().tolist()
    if n == 0:
        return float("nan"), float("nan")
    mean = s1 / n
    return mean, float(np.sqrt(max(s2 / n - mean * mean, 0.0)))


def _normalize(x, seg, sub, div, clip=None, zero_outside=False):
    lo, hi = clip if clip is not None else (0.0, 0.0)
    check(lib().cf_normalize(_f32(x), None if seg is None else _f32(seg), x.numel(), int(clip is not None), float(lo), float(hi), float(sub), float(div),
                             int(zero_outside), _stream()), "cf_normalize")


class GenericPreprocessor(object):
    """preprocessing.py:202-331 (test-time methods)."""

    resample_order_seg_test = 1

    def __init__(self, normalization_scheme_per_modality, use_nonzero_mask, transpose_forward, intensityproperties=None):
        self.transpose_forward = transpose_forward
        self.intensityproperties = intensityproperties
        self.normalization_scheme_per_modality = normalization_scheme_per_modality
        self.use_nonzero_mask = use_nonzero_mask
        self.resample_separate_z_anisotropy_threshold = RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD
        self.resample_order_data = 3
        self.resample_order_seg = 1

    def _target_spacing(self, target_spacing, original_spacing_transposed):
        return np.array(target_spacing, dtype=float)

    def resample_and_normalize(self, data, target_spacing, properties, seg=None, force_separate_z=None):
        """preprocessing.py:233-321.  data / seg already transposed by transpose_forward; properties are the un-transposed values."""
        was_numpy = not torch.is_tensor(data)
        original_spacing_transposed = np.array(properties["original_spacing"])[list(self.transpose_forward)]
        target_spacing = self._target_spacing(target_spacing, original_spacing_transposed)
        seg_dtype = None if seg is None else seg.dtype
        d = _to_dev(data)
        if self._remove_nans:
            d = d.clone() if d.data_ptr() == (data.data_ptr() if torch.is_tensor(data) else 0) else d
            check(lib().cf_nan_to_zero(_f32(d), d.numel(), _stream()), "cf_nan_to_zero")
        s = None if seg is None else _to_dev(seg)
        d, s = resample_patient(d, s, np.array(original_spacing_transposed), target_spacing, self.resample_order_data, self.resample_order_seg,
                                force_separate_z=force_separate_z, order_z_data=0, order_z_seg=0,
                                separate_z_anisotropy_threshold=self.resample_separate_z_anisotropy_threshold)
        d = d.contiguous()
        if s is not None:
            s = s.contiguous()
            s[s < -1] = 0
        properties["size_after_resampling"] = tuple(d[0].shape)
        properties["spacing_after_resampling"] = target_spacing
        assert len(self.normalization_scheme_per_modality) == len(d), "self.normalization_scheme_per_modality must have as many entries as data has modalities"
        assert len(self.use_nonzero_mask) == len(d), "self.use_nonzero_mask must have as many entries as data has modalities"
        for c in range(len(d)):
            scheme = self.normalization_scheme_per_modality[c]
            masked = bool(self.use_nonzero_mask[c])
            if scheme in ("CT", "CT2"):
                assert self.intensityproperties is not None, "ERROR: if there is a CT then we need intensity properties"
                ip = self.intensityproperties[c]
                lb, ub = ip["percentile_00_5"], ip["percentile_99_5"]
                if scheme == "CT":
                    mn, sd = ip["mean"], ip["sd"]
                else:
                    mn, sd = _moments(d[c], None, lb, ub)
                _normalize(d[c], s[-1] if masked else None, mn, sd, clip=(lb, ub), zero_outside=masked)
            elif scheme == "noNorm":
                pass
            else:
                mn, sd = _moments(d[c], s[-1] if masked else None)
                _normalize(d[c], s[-1] if masked else None, np.float32(mn), np.float32(sd) + np.float32(1e-8), zero_outside=masked)
        if was_numpy:
            return d.cpu().numpy(), None if s is None else s.cpu().numpy().astype(seg_dtype), properties
        return d, None if s is None else s.to(seg_dtype), properties

    _remove_nans = True

    def preprocess_test_case(self, data_files, target_spacing=None, seg_file=None, force_separate_z=None, need_seg=True):
        """preprocessing.py:323-331 -> (data float32 [C, ...], seg, properties), numpy like the reference.  The case goes to the device once
        and comes back once: crop, transpose, resampling and normalisation hand device tensors to each other (the numpy-in / numpy-out
        functions above cost two more 2 MB copies and a 4 MB int64 label map per frame, with a stream synchronisation each, which the
        file-level API paid under a running network batch -- profiles/r03_api_split.md).  need_seg=False skips the label map's read-back."""
        data, seg, properties = load_case_from_list_of_files(data_files, seg_file)
        data, seg, properties = ImageCropper.crop(_to_dev(data), properties, None if seg is None else _to_dev(seg))
        tf = (0, *[i + 1 for i in self.transpose_forward])
        if target_spacing is None:        # (plans without stages: the case keeps its own spacing)
            target_spacing = np.array(properties["original_spacing"], dtype=float)[list(self.transpose_forward)]
        d, s, properties = self.resample_and_normalize(data.permute(tf).contiguous(), target_spacing, properties, seg.permute(tf).contiguous(),
                                                       force_separate_z=force_separate_z)
        return d.cpu().numpy().astype(np.float32, copy=False), (s.cpu().numpy() if need_seg else None), properties

    def preprocess_arrays(self, data, seg, properties, target_spacing, force_separate_z=None):
        tf = (0, *[i + 1 for i in self.transpose_forward])
        data = data.transpose(tf)
        seg = seg.transpose(tf)
        data, seg, properties = self.resample_and_normalize(data, target_spacing, properties, seg, force_separate_z=force_separate_z)
        return data.astype(np.float32), seg, properties


class PreprocessorFor2D(GenericPreprocessor):
    """preprocessing.py:699-803: the first (slice) axis keeps its spacing; NaNs are not touched."""

    _remove_nans = False

    def _target_spacing(self, target_spacing, original_spacing_transposed):
        ts = np.array(target_spacing, dtype=float)
        ts[0] = original_spacing_transposed[0]
        return ts
