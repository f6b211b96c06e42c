#!/usr/bin/env python3
"""Generate tests/golden/preprocess_*.npz by RUNNING THE REFERENCE's preprocessing functions in the build container and pin
oracle/preprocess.py against them.

    cd /tmp && python /root/repo/tests/golden/make_golden_preprocess.py

skimage and batchgenerators are absent from the image, so `resize` / `resize_segmentation` inside the reference module are the
oracle's restatements of those two third-party functions (injected below, like pad_nd_image in make_golden.py): the pins cover the
reference's OWN code -- cropping.py, resample_patient / resample_data_or_seg control flow, the normalisation schemes -- and say
nothing about the two injected functions ("parity unpinned" for them, see oracle/preprocess.py).
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import preprocess as OP  # noqa: E402

REPORT = []


def pin(name, ref, ora, tol=0.0):
    ref, ora = np.asarray(ref), np.asarray(ora)
    assert ref.shape == ora.shape, (name, ref.shape, ora.shape)
    d = float(np.abs(ref.astype(np.float64) - ora.astype(np.float64)).max()) if ref.size else 0.0
    REPORT.append((name, d, tol))
    print("  oracle vs reference %-46s max|diff| = %.3e (tol %.1e)" % (name, d, tol))
    assert d <= tol, name


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote %-34s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def synthetic_volume(seed, shape=(2, 9, 44, 40)):
    """two modalities with a zero frame around the body, enclosed zero holes (filled by binary_fill_holes) and a zero channel
    that reaches the border (not filled)"""
    rng = np.random.default_rng(seed)
    C, Z, Y, X = shape
    d = (rng.normal(size=shape) * 120 + 400).astype(np.float32)
    d[:, :1] = 0
    d[:, :, :5] = 0
    d[:, :, -3:] = 0
    d[:, :, :, :4] = 0
    d[:, :, :, -6:] = 0
    d[:, 3:6, 15:22, 12:20] = 0          # enclosed hole
    d[:, 4, 30:33, :] = 0                # channel open to the border of that slice but enclosed in 3-D? (z neighbours are non-zero)
    d[:, 1:, 10:12, 30:] = 0             # open to the x border through the zero frame
    d[1, 2, 8, 8] = 0                    # zero in one modality only
    return d


def main():
    import nnunet.preprocessing.cropping as ref_crop
    import nnunet.preprocessing.preprocessing as ref_pp

    ref_pp.resize = OP.resize
    ref_pp.resize_segmentation = OP.resize_segmentation

    # ------------------------------------------------------------------ cropping.py
    data = synthetic_volume(1)
    seg_in = (np.random.default_rng(2).integers(0, 3, size=(1,) + data.shape[1:])).astype(np.float32)
    r_mask = ref_crop.create_nonzero_mask(data)
    pin("create_nonzero_mask", r_mask, OP.create_nonzero_mask(data))
    r = ref_crop.crop_to_nonzero(data.copy(), None, -1)
    o = OP.crop_to_nonzero(data.copy(), None, -1)
    pin("crop_to_nonzero data", r[0], o[0])
    pin("crop_to_nonzero created seg", r[1], o[1])
    pin("crop_to_nonzero bbox", np.array(r[2]), np.array(o[2]))
    r2 = ref_crop.crop_to_nonzero(data.copy(), seg_in.copy(), -1)
    o2 = OP.crop_to_nonzero(data.copy(), seg_in.copy(), -1)
    pin("crop_to_nonzero given seg", r2[1], o2[1])
    save("preprocess_crop", data=data, seg_in=seg_in, mask=r_mask, cropped=r[0], seg=r[1], bbox=np.array(r[2]), seg_given=r2[1])

    # ------------------------------------------------------------------ resample_patient (control flow of resample_data_or_seg)
    cropped, seg = r[0], r[1].astype(np.float32)
    cases = {
        # name: (original spacing, target spacing, force_separate_z)
        "aniso_z": ((10.0, 1.5625, 1.5625), (6.0, 1.25, 1.25), None),      # separate z (axis 0), z changes too
        "aniso_inplane": ((10.0, 1.5625, 1.5625), (10.0, 1.25, 1.4), None),  # separate z, slices only
        "iso": ((1.5, 1.4, 1.3), (1.2, 1.2, 1.2), None),                   # one 3-D resize
        "down": ((10.0, 1.25, 1.25), (10.0, 2.1, 1.9), None),               # down-sampling in-plane
    }
    out = {}
    for name, (osp, tsp, fsz) in cases.items():
        rd, rs = ref_pp.resample_patient(cropped.copy(), seg.copy(), np.array(osp), np.array(tsp), 3, 1, force_separate_z=fsz, order_z_data=0,
                                         order_z_seg=0)
        od, os_ = OP.resample_patient(cropped.copy(), seg.copy(), np.array(osp), np.array(tsp), 3, 1, force_separate_z=fsz, order_z_data=0,
                                      order_z_seg=0)
        pin("resample_patient[%s] data" % name, rd, od)
        pin("resample_patient[%s] seg" % name, rs, os_)
        out[name + "_data"], out[name + "_seg"] = rd, rs
        out[name + "_spacing"] = np.array([osp, tsp])
    # the export direction: order 1 data with order_z 0, and a segmentation of order 0
    rd = ref_pp.resample_data_or_seg(cropped.copy(), (12, 50, 41), False, [0], 1, True, order_z=0)
    pin("resample_data_or_seg order1/z0", rd, OP.resample_data_or_seg(cropped.copy(), (12, 50, 41), False, [0], 1, True, order_z=0))
    out["lin_data"] = rd
    save("preprocess_resample", cropped=cropped, seg=seg, **out)

    # ------------------------------------------------------------------ the preprocessors' resample_and_normalize
    ip = {0: {"mean": 380.0, "sd": 110.0, "percentile_00_5": 120.0, "percentile_99_5": 650.0},
          1: {"mean": 400.0, "sd": 125.0, "percentile_00_5": 100.0, "percentile_99_5": 700.0}}
    norm = {}
    configs = {
        "nonct": (ref_pp.GenericPreprocessor, False, {0: "nonCT", 1: "nonCT"}, {0: False, 1: False}),
        "nonct_mask": (ref_pp.GenericPreprocessor, False, {0: "nonCT", 1: "nonCT"}, {0: True, 1: True}),
        "ct": (ref_pp.GenericPreprocessor, False, {0: "CT", 1: "CT2"}, {0: True, 1: False}),
        "nonorm2d": (ref_pp.PreprocessorFor2D, True, {0: "noNorm", 1: "nonCT"}, {0: False, 1: True}),
    }
    props0 = {"original_spacing": np.array([10.0, 1.5625, 1.5625])}
    tf = [0, 1, 2]
    for name, (cls, two_d, schemes, use_mask) in configs.items():
        pre = cls(schemes, use_mask, tf, ip)
        seg_t = r[1].copy()
        rd, rs, rp = pre.resample_and_normalize(cropped.copy(), np.array([8.0, 1.25, 1.25]), copy.deepcopy(props0), seg_t, force_separate_z=None)
        od, os_, op = OP.resample_and_normalize(cropped.copy(), np.array([8.0, 1.25, 1.25]), copy.deepcopy(props0), r[1].copy(), tf, schemes, use_mask,
                                                ip, None, two_d)
        pin("resample_and_normalize[%s] data" % name, rd, od, 0.0)
        pin("resample_and_normalize[%s] seg" % name, rs, os_, 0.0)
        assert tuple(rp["size_after_resampling"]) == tuple(op["size_after_resampling"])
        norm[name + "_data"], norm[name + "_seg"] = rd.astype(np.float32), rs
    save("preprocess_normalize", cropped=cropped, seg=r[1], **norm)

    print("\nall %d preprocessing pins within tolerance" % len(REPORT))
    with open(os.path.join(HERE, "PIN_REPORT_preprocess.txt"), "w") as f:
        f.write("oracle/preprocess.py vs the reference's cropping.py / preprocessing.py (make_golden_preprocess.py, build container);\n")
        f.write("`resize` / `resize_segmentation` (skimage, batchgenerators: absent) are the oracle's restatements injected into the reference module\n")
        for n, d, t in REPORT:
            f.write("%-56s max|diff| %.3e  tol %.1e\n" % (n, d, t))


if __name__ == "__main__":
    main()
