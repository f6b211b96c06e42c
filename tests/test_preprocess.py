"""Test-time preprocessing: the oracle against the reference's golden vectors on the CPU, the device path
(cineflow.preprocessing over the C ABI) against the oracle and the same vectors on the GPU.

Tolerances: masks, bounding boxes and label maps must be identical; resampled intensities are computed in fp64 on both sides and
rounded to fp32 (values ~ 400, one fp32 ulp = 3e-5), so 1e-4 abs; normalised data is O(1) fp32 arithmetic, 2e-5 abs."""
import copy

import numpy as np
import pytest

CASES = {"aniso_z": None, "aniso_inplane": None, "iso": None, "down": None}
IP = {0: {"mean": 380.0, "sd": 110.0, "percentile_00_5": 120.0, "percentile_99_5": 650.0},
      1: {"mean": 400.0, "sd": 125.0, "percentile_00_5": 100.0, "percentile_99_5": 700.0}}
NORM = {
    "nonct": (False, {0: "nonCT", 1: "nonCT"}, {0: False, 1: False}),
    "nonct_mask": (False, {0: "nonCT", 1: "nonCT"}, {0: True, 1: True}),
    "ct": (False, {0: "CT", 1: "CT2"}, {0: True, 1: False}),
    "nonorm2d": (True, {0: "noNorm", 1: "nonCT"}, {0: False, 1: True}),
}


# ------------------------------------------------------------------------------------------------ oracle vs golden (CPU)
def test_oracle_crop_golden(golden):
    from oracle import preprocess as OP
    g = golden("preprocess_crop")
    assert np.array_equal(OP.create_nonzero_mask(g["data"]), g["mask"])
    assert g["mask"].sum() > (g["data"] != 0).any(0).sum(), "the fixture must contain filled holes"
    d, s, bbox = OP.crop_to_nonzero(g["data"].copy(), None, -1)
    assert np.array_equal(d, g["cropped"]) and np.array_equal(s, g["seg"]) and np.array_equal(np.array(bbox), g["bbox"])
    assert np.array_equal(OP.crop_to_nonzero(g["data"].copy(), g["seg_in"].copy(), -1)[1], g["seg_given"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_resample_golden(golden, name):
    from oracle import preprocess as OP
    g = golden("preprocess_resample")
    osp, tsp = g[name + "_spacing"]
    d, s = OP.resample_patient(g["cropped"].copy(), g["seg"].copy(), osp, tsp, 3, 1, force_separate_z=None, order_z_data=0, order_z_seg=0)
    assert np.array_equal(d, g[name + "_data"]) and np.array_equal(s, g[name + "_seg"])


@pytest.mark.parametrize("name", sorted(NORM))
def test_oracle_normalize_golden(golden, name):
    from oracle import preprocess as OP
    g = golden("preprocess_normalize")
    two_d, schemes, use_mask = NORM[name]
    d, s, props = OP.resample_and_normalize(g["cropped"].copy(), np.array([8.0, 1.25, 1.25]), {"original_spacing": np.array([10.0, 1.5625, 1.5625])},
                                            g["seg"].copy(), [0, 1, 2], schemes, use_mask, IP, None, two_d)
    assert np.array_equal(d.astype(np.float32), g[name + "_data"]) and np.array_equal(s, g[name + "_seg"])
    assert tuple(props["size_after_resampling"]) == g[name + "_data"].shape[1:]


def test_oracle_resize_is_scipy_zoom():
    """the restated third-party `resize` (parity unpinned) against the scipy call it is defined by, and its edge cases"""
    from scipy import ndimage as ndi
    from oracle import preprocess as OP
    x = np.random.default_rng(0).normal(size=(7, 11))
    assert np.array_equal(OP.resize(x, (7, 11), 3), x)
    z = ndi.zoom(x, [2, 13 / 11], order=3, mode="nearest", grid_mode=True)
    assert np.array_equal(OP.resize(x, (14, 13), 3), np.clip(z, x.min(), x.max()))
    lab = np.array([[0, 0, 2], [1, 1, 2]])
    assert set(np.unique(OP.resize_segmentation(lab, (5, 7), 1))) <= {0, 1, 2}


# ------------------------------------------------------------------------------------------------ device vs oracle / golden
@pytest.mark.gpu
def test_crop_to_nonzero_device(dev, golden):
    import torch
    from cineflow import preprocessing as P
    g = golden("preprocess_crop")
    assert np.array_equal(P.create_nonzero_mask(g["data"]), g["mask"])
    assert np.array_equal(P.create_nonzero_mask(g["data"][:, 4]), __import__("oracle.preprocess", fromlist=["x"]).create_nonzero_mask(g["data"][:, 4]))
    assert P.get_bbox_from_mask(g["mask"]) == g["bbox"].tolist()
    d, s, bbox = P.crop_to_nonzero(g["data"].copy(), None, -1)
    assert np.array_equal(d, g["cropped"]) and np.array_equal(s, g["seg"]) and bbox == g["bbox"].tolist()
    assert np.array_equal(P.crop_to_nonzero(g["data"].copy(), g["seg_in"].copy(), -1)[1], g["seg_given"])
    # device tensors stay on the device
    dt, st, _ = P.crop_to_nonzero(torch.from_numpy(g["data"]).to(dev))
    assert dt.is_cuda and st.is_cuda and np.array_equal(dt.cpu().numpy(), g["cropped"])
    with pytest.raises(ValueError):
        P.get_bbox_from_mask(np.zeros((3, 4, 5), bool))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_resample_patient_device(dev, golden, name):
    from cineflow import preprocessing as P
    g = golden("preprocess_resample")
    osp, tsp = g[name + "_spacing"]
    d, s = P.resample_patient(g["cropped"].copy(), g["seg"].copy(), osp, tsp, 3, 1, force_separate_z=None, order_z_data=0, order_z_seg=0)
    assert d.shape == g[name + "_data"].shape and d.dtype == np.float32
    assert float(np.abs(d.astype(np.float64) - g[name + "_data"]).max()) <= 1e-4
    assert np.array_equal(s, g[name + "_seg"])


@pytest.mark.gpu
def test_resample_orders_device(dev, golden):
    from cineflow import preprocessing as P
    from oracle import preprocess as OP
    g = golden("preprocess_resample")
    lin = P.resample_data_or_seg(g["cropped"].copy(), (12, 50, 41), False, [0], 1, True, order_z=0)
    assert float(np.abs(lin - g["lin_data"]).max()) <= 1e-4
    # a single line, a single slice and up/down factors that are not dyadic: against the oracle
    rng = np.random.default_rng(5)
    for shape, new in (((1, 1, 1, 9), (1, 1, 23)), ((2, 1, 13, 9), (1, 30, 7)), ((1, 3, 5, 4), (7, 9, 11))):
        x = (rng.normal(size=shape) * 50).astype(np.float32)
        got = P.resample_data_or_seg(x.copy(), new, False, None, 3, False)
        ref = OP.resample_data_or_seg(x.copy(), new, False, None, 3, False)
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 2e-5, (shape, new)
    with pytest.raises(NotImplementedError):
        P.resample_data_or_seg(g["cropped"].copy(), (12, 50, 41), False, [0], 2, True)
    same = g["cropped"].copy()
    assert P.resample_data_or_seg(same, same.shape[1:], False) is same


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(NORM))
def test_resample_and_normalize_device(dev, golden, name):
    from cineflow import preprocessing as P
    g = golden("preprocess_normalize")
    two_d, schemes, use_mask = NORM[name]
    cls = P.PreprocessorFor2D if two_d else P.GenericPreprocessor
    pre = cls(schemes, use_mask, [0, 1, 2], IP)
    props = {"original_spacing": np.array([10.0, 1.5625, 1.5625])}
    d, s, props = pre.resample_and_normalize(g["cropped"].copy(), np.array([8.0, 1.25, 1.25]), props, g["seg"].copy())
    ref = g[name + "_data"]
    assert d.shape == ref.shape and tuple(props["size_after_resampling"]) == ref.shape[1:]
    tol = 1e-4 if name == "nonorm2d" else 2e-5          # the un-normalised channel keeps its ~400 magnitude
    assert float(np.abs(d.astype(np.float64) - ref).max()) <= tol
    assert np.array_equal(s, g[name + "_seg"])


@pytest.mark.gpu
def test_preprocess_test_case_files(dev, golden, tmp_path):
    """the file-level entry point trainer.preprocess_patient uses: NIfTI in -> (data, seg, properties) like the reference's
    GenericPreprocessor.preprocess_test_case, checked against the oracle's crop + transpose + resample + normalise chain"""
    from cineflow import preprocessing as P
    from cineflow.nifti import write_nifti
    from oracle import preprocess as OP
    g = golden("preprocess_crop")
    spacing_xyz = (1.5625, 1.5625, 10.0)
    files = []
    for c in range(2):
        f = str(tmp_path / ("case_%04d.nii.gz" % c))
        write_nifti(f, g["data"][c], spacing=spacing_xyz)
        files.append(f)
    schemes, use_mask = {0: "nonCT", 1: "nonCT"}, {0: False, 1: True}
    d, s, props = P.PreprocessorFor2D(schemes, use_mask, [0, 1, 2]).preprocess_test_case(files, np.array([999.0, 1.25, 1.25]))
    od, os_, op = OP.preprocess_arrays(g["data"].copy(), {"original_spacing": np.array(spacing_xyz)[::-1]}, np.array([999.0, 1.25, 1.25]), [0, 1, 2],
                                       schemes, use_mask, None, None, True)
    assert d.dtype == np.float32 and d.shape == od.shape
    assert float(np.abs(d - od).max()) <= 2e-5 and np.array_equal(s, os_)
    assert props["crop_bbox"] == op["crop_bbox"] and tuple(props["size_after_cropping"]) == tuple(op["size_after_cropping"])
    assert np.allclose(props["original_spacing"], np.array(spacing_xyz)[::-1]) and list(props["classes"]) == [-1, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("tf", [[0, 1, 2], [2, 0, 1]])
def test_preprocess_test_case_one_round_trip_equals_function_chain(dev, golden, tmp_path, tf):
    """preprocess_test_case keeps the case on the device between crop, transpose, resampling and normalisation; the reference-shaped numpy-in /
    numpy-out functions (crop_from_list_of_files -> preprocess_arrays) go through the host between them: same kernels, so identical arrays"""
    from cineflow import preprocessing as P
    from cineflow.nifti import write_nifti
    g = golden("preprocess_crop")
    files = []
    for c in range(2):
        f = str(tmp_path / ("case_%04d.nii.gz" % c))
        write_nifti(f, g["data"][c], spacing=(1.5625, 1.5625, 10.0))
        files.append(f)
    pre = P.GenericPreprocessor({0: "nonCT", 1: "nonCT"}, {0: False, 1: True}, tf)
    target = np.array([5.0, 1.25, 1.25])[tf] if tf != [0, 1, 2] else np.array([10.0, 1.25, 1.25])
    d1, s1, p1 = pre.preprocess_test_case(files, target)
    data, seg, props = P.ImageCropper.crop_from_list_of_files(files)
    d2, s2, p2 = pre.preprocess_arrays(data, seg, props, target)
    assert d1.dtype == d2.dtype == np.float32 and np.array_equal(d1, d2) and np.array_equal(s1, s2)
    for k in ("crop_bbox", "size_after_cropping", "size_after_resampling"):
        assert tuple(map(tuple, p1[k])) == tuple(map(tuple, p2[k])) if k == "crop_bbox" else tuple(p1[k]) == tuple(p2[k])
    assert list(p1["classes"]) == list(p2["classes"])
    d3, s3, _ = pre.preprocess_test_case(files, target, need_seg=False)
    assert s3 is None and np.array_equal(d3, d1)
    d4, _, _ = pre.preprocess_test_case(files)          # no target spacing: the case keeps its own
    assert d4.shape[1:] == tuple(np.array(p1["size_after_cropping"])[tf])
