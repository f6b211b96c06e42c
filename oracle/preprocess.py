"""Oracle restatement of the reference's test-time preprocessing, the step right before the hot path.

Test infrastructure only -- see oracle/__init__.py.  All `file:line` citations are relative to /root/reference.

Pinned against the reference (tests/golden/make_golden.py -> preprocess_*.npz): crop_to_nonzero, resample_data_or_seg's own
control flow (per-slice loop, z pass, label handling, dtype casts) and resample_and_normalize's normalisation schemes.
NOT pinned: `resize` and `resize_segmentation` themselves -- skimage and batchgenerators are absent from this image, so the two
functions are restated from their published algorithms and injected into the reference module when the pins are generated.
"""
import numpy as np

RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3  # nnunet/configuration.py


# --------------------------------------------------------------------------- third-party restatements (parity unpinned)
def resize(image, output_shape, order=1, mode="edge", clip=True, anti_aliasing=False, **_):
    """skimage.transform.resize (scikit-image >= 0.19, anti_aliasing=False): scipy.ndimage.zoom with grid_mode=True, i.e. sampling
    at src = (dst + 0.5) * n_src / n_dst - 0.5; mode 'edge' = scipy 'nearest' (order >= 2 prefilters the 12-sample edge-padded
    image); the result is clipped to the input range."""
    from scipy import ndimage as ndi
    assert mode == "edge" and not anti_aliasing
    image = np.asarray(image, dtype=float)
    output_shape = tuple(int(v) for v in output_shape)
    if image.shape == output_shape:
        return image.copy()
    zoom = [o / float(i) for o, i in zip(output_shape, image.shape)]
    out = ndi.zoom(image, zoom, order=order, mode="nearest", grid_mode=True)
    assert out.shape == output_shape
    if clip and order > 1:
        out = np.clip(out, image.min(), image.max())
    return out


def resize_segmentation(segmentation, new_shape, order=3):
    """batchgenerators.augmentations.utils.resize_segmentation (batchgenerators >= 0.23, nnU-Net's pin): order 0 resizes the label
    map directly; any other order resizes each label's indicator and assigns the label where it reaches 0.5, labels in ascending
    order so a later label overwrites an earlier one."""
    tpe = segmentation.dtype
    unique_labels = np.unique(segmentation)
    assert len(segmentation.shape) == len(new_shape)
    if order == 0:
        return resize(segmentation.astype(float), new_shape, order, mode="edge", clip=True, anti_aliasing=False).astype(tpe)
    reshaped = np.zeros(new_shape, dtype=segmentation.dtype)
    for c in unique_labels:
        mask = segmentation == c
        reshaped_multihot = resize(mask.astype(float), new_shape, order, mode="edge", clip=True, anti_aliasing=False)
        reshaped[reshaped_multihot >= 0.5] = c
    return reshaped


# --------------------------------------------------------------------------- cropping
def create_nonzero_mask(data):
    """nnunet/preprocessing/cropping.py:25-32."""
    from scipy.ndimage import binary_fill_holes
    assert data.ndim in (3, 4)
    nonzero_mask = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        nonzero_mask |= data[c] != 0
    return binary_fill_holes(nonzero_mask)


def get_bbox_from_mask(mask, outside_value=0):
    """cropping.py:47-55."""
    co = np.where(mask != outside_value)
    return [[int(np.min(co[a])), int(np.max(co[a])) + 1] for a in range(3)]


def crop_to_bbox(image, bbox):
    """cropping.py:58-61."""
    return image[bbox[0][0]:bbox[0][1], bbox[1][0]:bbox[1][1], bbox[2][0]:bbox[2][1]]


def crop_to_nonzero(data, seg=None, nonzero_label=-1):
    """cropping.py:104-137: crop every channel to the bounding box of the hole-filled non-zero mask; voxels outside the mask get
    `nonzero_label` in the segmentation (which is created when absent)."""
    nonzero_mask = create_nonzero_mask(data)
    bbox = get_bbox_from_mask(nonzero_mask, 0)
    data = np.vstack([crop_to_bbox(data[c], bbox)[None] for c in range(data.shape[0])])
    if seg is not None:
        seg = np.vstack([crop_to_bbox(seg[c], bbox)[None] for c in range(seg.shape[0])])
    nonzero_mask = crop_to_bbox(nonzero_mask, bbox)[None]
    if seg is not None:
        seg[(seg == 0) & (nonzero_mask == 0)] = nonzero_label
    else:
        nonzero_mask = nonzero_mask.astype(int)
        nonzero_mask[nonzero_mask == 0] = nonzero_label
        nonzero_mask[nonzero_mask > 0] = 0
        seg = nonzero_mask
    return data, seg, bbox


# --------------------------------------------------------------------------- resampling
def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """preprocessing.py:30-32."""
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    """preprocessing.py:35-37."""
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def resample_data_or_seg(data, new_shape, is_seg, axis=None, order=3, do_separate_z=False, order_z=0):
    """preprocessing.py:111-200, control flow kept: with do_separate_z every slice along `axis` is resized in-plane (cast back to
    the input dtype), then one map_coordinates pass of order `order_z` runs along `axis` (per-label with rounding for
    segmentations of order_z > 0); otherwise one n-d resize per channel."""
    from scipy.ndimage import map_coordinates
    assert data.ndim == 4 and len(new_shape) == 3
    if is_seg:
        resize_fn, kwargs = resize_segmentation, {}
    else:
        resize_fn, kwargs = resize, {"mode": "edge", "anti_aliasing": False}
    dtype_data = data.dtype
    shape = np.array(data[0].shape)
    new_shape = np.array(new_shape)
    if not np.any(shape != new_shape):
        return data
    data = data.astype(float)
    if not do_separate_z:
        return np.vstack([resize_fn(data[c], new_shape, order, **kwargs)[None].astype(dtype_data) for c in range(data.shape[0])]).astype(dtype_data)
    assert len(axis) == 1, "only one anisotropic axis supported"
    axis = int(axis[0])
    new_shape_2d = new_shape[[a for a in range(3) if a != axis]]
    final = []
    for c in range(data.shape[0]):
        slices = []
        for s in range(shape[axis]):
            sl = [slice(None)] * 3
            sl[axis] = s
            slices.append(resize_fn(data[c][tuple(sl)], new_shape_2d, order, **kwargs).astype(dtype_data))
        reshaped_data = np.stack(slices, axis)
        if shape[axis] != new_shape[axis]:
            rows, cols, dim = (int(v) for v in new_shape)
            orig_rows, orig_cols, orig_dim = reshaped_data.shape
            map_rows, map_cols, map_dims = np.mgrid[:rows, :cols, :dim]
            coord_map = np.array([float(orig_rows) / rows * (map_rows + 0.5) - 0.5, float(orig_cols) / cols * (map_cols + 0.5) - 0.5,
                                  float(orig_dim) / dim * (map_dims + 0.5) - 0.5])
            if not is_seg or order_z == 0:
                final.append(map_coordinates(reshaped_data, coord_map, order=order_z, mode="nearest")[None].astype(dtype_data))
            else:
                reshaped = np.zeros(new_shape, dtype=dtype_data)
                for cl in np.unique(reshaped_data):
                    multihot = np.round(map_coordinates((reshaped_data == cl).astype(float), coord_map, order=order_z, mode="nearest"))
                    reshaped[multihot > 0.5] = cl
                final.append(reshaped[None].astype(dtype_data))
        else:
            final.append(reshaped_data[None].astype(dtype_data))
    return np.vstack(final).astype(dtype_data)


def resample_patient(data, seg, original_spacing, target_spacing, order_data=3, order_seg=0, force_separate_z=False, order_z_data=0,
                     order_z_seg=0, separate_z_anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """preprocessing.py:40-108."""
    assert not (data is None and seg is None)
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
        do_separate_z = False
    data_r = resample_data_or_seg(data, new_shape, False, axis, order_data, do_separate_z, order_z=order_z_data) if data is not None else None
    seg_r = resample_data_or_seg(seg, new_shape, True, axis, order_seg, do_separate_z, order_z=order_z_seg) if seg is not None else None
    return data_r, seg_r


# --------------------------------------------------------------------------- the preprocessor
def resample_and_normalize(data, target_spacing, properties, seg, transpose_forward, normalization_schemes, use_nonzero_mask,
                           intensityproperties=None, force_separate_z=None, two_d=False):
    """GenericPreprocessor.resample_and_normalize (preprocessing.py:233-321) / PreprocessorFor2D's (:732-803; two_d=True keeps the
    first axis' spacing and resamples segmentations with order 1)."""
    original_spacing_transposed = np.array(properties["original_spacing"])[list(transpose_forward)]
    target_spacing = np.array(target_spacing, dtype=float)
    if two_d:
        target_spacing[0] = original_spacing_transposed[0]
    else:
        data[np.isnan(data)] = 0
    data, seg = resample_patient(data, seg, np.array(original_spacing_transposed), target_spacing, 3, 1, force_separate_z=force_separate_z,
                                 order_z_data=0, order_z_seg=0)
    if seg is not None:
        seg[seg < -1] = 0
    properties["size_after_resampling"] = data[0].shape
    properties["spacing_after_resampling"] = target_spacing
    for c in range(len(data)):
        scheme = normalization_schemes[c]
        if scheme == "CT":
            ip = intensityproperties[c]
            data[c] = np.clip(data[c], ip["percentile_00_5"], ip["percentile_99_5"])
            data[c] = (data[c] - ip["mean"]) / ip["sd"]
            if use_nonzero_mask[c]:
                data[c][seg[-1] < 0] = 0
        elif scheme == "CT2":
            ip = intensityproperties[c]
            lb, ub = ip["percentile_00_5"], ip["percentile_99_5"]
            mask = (data[c] > lb) & (data[c] < ub)
            data[c] = np.clip(data[c], lb, ub)
            mn, sd = data[c][mask].mean(), data[c][mask].std()
            data[c] = (data[c] - mn) / sd
            if use_nonzero_mask[c]:
                data[c][seg[-1] < 0] = 0
        elif scheme == "noNorm":
            pass
        elif use_nonzero_mask[c]:
            mask = seg[-1] >= 0
            data[c][mask] = (data[c][mask] - data[c][mask].mean()) / (data[c][mask].std() + 1e-8)
            data[c][mask == 0] = 0
        else:
            data[c] = (data[c] - data[c].mean()) / (data[c].std() + 1e-8)
    return data, seg, properties


def preprocess_arrays(data, properties, target_spacing, transpose_forward, normalization_schemes, use_nonzero_mask, intensityproperties=None,
                      force_separate_z=None, two_d=False):
    """ImageCropper.crop (cropping.py:161-173) + GenericPreprocessor.preprocess_test_case (preprocessing.py:323-331) on an array that is
    already loaded: (data[C,Z,Y,X] float32, properties with original_spacing) -> (data float32, seg, properties)."""
    data, seg, bbox = crop_to_nonzero(data.astype(np.float32), None, nonzero_label=-1)
    properties["crop_bbox"] = bbox
    properties["classes"] = np.unique(seg)
    seg[seg < -1] = 0
    properties["size_after_cropping"] = data[0].shape
    tf = [0] + [i + 1 for i in transpose_forward]
    data, seg = data.transpose(tf), seg.transpose(tf)
    data, seg, properties = resample_and_normalize(data, target_spacing, properties, seg, transpose_forward, normalization_schemes,
                                                   use_nonzero_mask, intensityproperties, force_separate_z, two_d)
    return data.astype(np.float32), seg, properties
