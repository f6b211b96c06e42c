"""The `voxelmorph_saver_*` layout of the reference, produced and consumed on the HIP path.

`nnunet/voxelmorph_saver_Lib.py` (and its `_ACDC` / `_Lib_45` siblings) turns the CROP-SPACE predictions of a flow model

    <pred>/Raw/Registered/<patient>/<case>.nii.gz      propagated labels   [H, W, D]        (nibabel axis order)
    <pred>/Raw/Segmentation/<patient>/<case>.npz       'seg'  softmax      [C, H, W, D]     (also the ED frame, which has no Registered file)
    <pred>/Raw/Flow/<patient>/<case>.npz               'flow'              [H, W, D, 2]
    <pkl_path>/<case>.pkl                              the case's nnU-Net properties + 'padding_need' [4, D] + 'voxelmorph_size_before'

into native-geometry files

    <pred>/Postprocessed/{Flow,Registered,Segmentation}/<patient>/<case>.{npz,nii.gz,nii.gz}
    <pred>/Postprocessed/{Registered,Segmentation}/<patient>/temp_allClasses/<case>.nii.gz      (largest-component filter)

which `compute_metrics.py:51-56` (glob `<dir>/<patient>/temp_allClasses/*.gz`) and `compute_jacobian.py:128-139` (stack of
`Flow/<patient>/*.npz['flow']` -> [T, H, W, D, 2]) read.  Per file (voxelmorph_saver_Lib.py:193-265): every slice is un-cropped
with its own `padding_need[:, d]` (Processor.uncrop_no_registration, processor.py:178-186), the stack is centre padded / cropped to
`voxelmorph_size_before` (monai ResizeWithPadOrCrop), axes go back through `transpose_backward`, and
`save_segmentation_nifti_from_softmax` resamples to the size before resampling, rescales the flow and places everything in the crop
box; `determine_postprocessing_custom` (connected_components.py:574-596) then keeps the largest component of classes 1, 2, 3.

`write_raw` is the producer side: it writes that `Raw/` tree and the `.pkl` files from the outputs of
`CineTrainer.predict_preprocessed_data_return_seg_and_softmax_flow` (`cineflow.predict.set_voxelmorph_raw(pred_path, pkl_path)`, CLI
`--voxelmorph_raw`).  The `.pkl` files hold plain Python / numpy values only and are written with `pickle.dump`.  The consumer reads
them with `_PlainUnpickler`, whose `find_class` admits OrderedDict and numpy's array / dtype / scalar reconstructors and nothing else:
a `.pkl` tree written by the reference pipeline loads (its properties are exactly such values), a pickle that names any other
global raises `pickle.UnpicklingError` instead of importing or calling it.
monai and nibabel are absent here: ResizeWithPadOrCrop is restated from its documented centre rule (parity unpinned), NIfTI goes
through cineflow.nifti in nibabel's (i, j, k) axis order.
"""
import os
import pickle
import shutil

import numpy as np
import torch

from . import ops
from .inference import Processor
from .nifti import read_nifti, write_nifti
from .predict import load_remove_save, save_segmentation_nifti_from_softmax, subfiles

join = os.path.join


# ------------------------------------------------------------------------------------------------ small pieces
class _PlainUnpickler(pickle.Unpickler):
    """pickle.Unpickler that can only rebuild plain containers and numpy values (what nnU-Net property dicts hold).  `pkl_path` is a free
    CLI argument and its natural input is a tree another pipeline wrote, so no global outside this list is ever resolved."""

    _ALLOWED = {
        ("collections", "OrderedDict"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing to load global %s.%s from a properties .pkl (plain containers and numpy values only)" % (module, name))


def load_plain_pickle(path):
    with open(path, "rb") as f:
        return _PlainUnpickler(f).load()


def delete_if_exist(folder_name):
    """voxelmorph_saver_Lib.py:279-282."""
    if os.path.isdir(folder_name):
        shutil.rmtree(folder_name)


def resize_with_pad_or_crop(arr, spatial_size):
    """monai.transforms.ResizeWithPadOrCrop(spatial_size)(arr) for a channel-first array [C, ...]: symmetric zero padding
    (before = total // 2) up to the target, then a centre crop (start = size // 2 - target // 2), axis by axis."""
    a = np.asarray(arr)
    assert a.ndim == len(spatial_size) + 1
    pads = [(0, 0)]
    for n, t in zip(a.shape[1:], spatial_size):
        tot = max(int(t) - n, 0)
        pads.append((tot // 2, tot - tot // 2))
    if any(p != (0, 0) for p in pads):
        a = np.pad(a, pads, mode="constant")
    sl = [slice(None)]
    for n, t in zip(a.shape[1:], spatial_size):
        start = max(n // 2 - int(t) // 2, 0)
        sl.append(slice(start, start + int(t)))
    return a[tuple(sl)]


def _nib_write(path, arr_hwd, dtype=np.uint8):
    """write an array in nibabel's axis order (first axis fastest in the file) with identity geometry"""
    write_nifti(path, np.ascontiguousarray(np.asarray(arr_hwd).astype(dtype).transpose(2, 1, 0)))


def _nib_read(path):
    """nib.load(path).get_fdata(): the array in file axis order [i, j, k]"""
    a, _ = read_nifti(path)
    return a.transpose(2, 1, 0).astype(np.float64)


def uncrop_stack(arr, padding_need, processor, device):
    """arr [C, H, W, D] in crop space -> [C, image, image, D]: slice d is zero padded by padding_need[:, d] = (left, right, top, bottom)
    (voxelmorph_saver_Lib.py:212-222; the pad itself is cf_pad2d on the device)."""
    C, H, W, D = arr.shape
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, np.float32).transpose(3, 0, 1, 2))).to(device)      # [D, C, H, W]
    out = [processor.uncrop_no_registration(t[d].contiguous(), [int(v) for v in np.asarray(padding_need)[:, d]]) for d in range(D)]
    return torch.stack(out, dim=-1).cpu().numpy()


# ------------------------------------------------------------------------------------------------ producer: Raw/ + pkl
def write_raw(pred_path, pkl_path, patient_name, case_names, softmax, flow, registered, property_list, padding_need, size_before,
              ed_position=0):
    """Write one patient's crop-space predictions in the layout voxelmorph_saver_* reads (module docstring).

    case_names[t]     file stem of frame t (e.g. 'patient001_frame01'), ED first as predict_cases orders them
    softmax[t]        [K, D, h, w] crop-space class probabilities;  flow[t] [2, D, h, w];  registered[t] [D, h, w] or [1, D, h, w]
    property_list[t]  the frame's nnU-Net properties dict (plain values);  padding_need [4, D];  size_before = [H, W, D] of the
                      volume before the voxelmorph crop.  The ED frame (index ed_position) gets a Segmentation file only."""
    sub = {k: join(pred_path, "Raw", k, patient_name) for k in ("Registered", "Segmentation", "Flow")}
    for d in sub.values():
        os.makedirs(d, exist_ok=True)
    os.makedirs(pkl_path, exist_ok=True)
    padding_need = np.asarray(padding_need, dtype=np.int64)
    assert padding_need.ndim == 2 and padding_need.shape[0] == 4, "padding_need must be [4, D]"
    for t, name in enumerate(case_names):
        sm = np.asarray(softmax[t], np.float32)                                    # [K, D, h, w] -> [K, h, w, D]
        np.savez(join(sub["Segmentation"], name + ".npz"), seg=sm.transpose(0, 2, 3, 1))
        props = dict(property_list[t])
        props["padding_need"] = padding_need
        props["voxelmorph_size_before"] = [int(v) for v in size_before]
        with open(join(pkl_path, name + ".pkl"), "wb") as f:
            pickle.dump(props, f)
        if t == ed_position:
            continue
        fl = np.asarray(flow[t], np.float32)                                       # [2, D, h, w] -> [h, w, D, 2]
        np.savez(join(sub["Flow"], name + ".npz"), flow=fl.transpose(2, 3, 1, 0))
        rg = np.asarray(registered[t])
        rg = rg[0] if rg.ndim == 4 else rg                                         # [D, h, w] -> [h, w, D]
        _nib_write(join(sub["Registered"], name + ".nii.gz"), rg.transpose(1, 2, 0))
    return sub


# ------------------------------------------------------------------------------------------------ consumer: Postprocessed/
def determine_postprocessing_custom(base, raw_subfolder_name="validation_raw", temp_folder="temp", final_subf_name="validation_final",
                                    processes=1, dice_threshold=0, debug=True, advanced_postprocessing=False,
                                    pp_filename="postprocessing.json", nb_threads=1, log_function=print):
    """connected_components.py:574-616: every .nii.gz of <base>/<raw_subfolder_name> through load_remove_save with classes
    [1, 2, 3] into <base>/<temp_folder>_allClasses (the largest-component kernels of cineflow.ops)."""
    classes = [1, 2, 3]
    folder_all_classes_as_fg = join(base, temp_folder + "_allClasses")
    if os.path.isdir(folder_all_classes_as_fg):
        shutil.rmtree(folder_all_classes_as_fg)
    fnames = subfiles(join(base, raw_subfolder_name) if raw_subfolder_name else base, suffix=".nii.gz", join_=False)
    os.makedirs(folder_all_classes_as_fg, exist_ok=True)
    for f in fnames:
        load_remove_save(join(base, raw_subfolder_name, f) if raw_subfolder_name else join(base, f), join(folder_all_classes_as_fg, f), classes)
    log_function("done")
    return folder_all_classes_as_fg


def build_cropping_network(cropper_config, image_size, window_size, weights=None, device=None):
    """voxelmorph_saver_Lib.py:340-345: `read_config(adversarial_acdc.yaml)` -> `build_2d_model(..., num_classes=2, processor=None)` ->
    `load_state_dict(torch.load(<cropper weights>/model_final_checkpoint.model)['state_dict'], strict=True)`.  cropper_config: the YAML's
    file name or its mapping; weights: a `.model` file holding {'state_dict': {name: tensor}} (read with torch.load(weights_only=True): a
    tensor dict loads, a pickled trainer object is refused) or such a dict itself."""
    from . import config as C
    cfg = C.read_config(cropper_config, False, False) if isinstance(cropper_config, str) else cropper_config
    net = C.build_2d_model(cfg, conv_layer=None, norm=None, log_function=None, image_size=image_size, window_size=window_size, middle=False,
                           num_classes=2, processor=None)
    if weights is not None:
        sd = torch.load(weights, map_location="cpu", weights_only=True) if isinstance(weights, str) else weights
        sd = sd["state_dict"] if "state_dict" in sd else sd
        net.load_state_dict(sd, device or torch.device("cuda", torch.cuda.current_device()), strict=True)
    return net


class Saver:
    """The module-level state of voxelmorph_saver_Lib.py's __main__ (processor, plans entries) as an object.  `cropping_network`: the
    2-class MTLmodel the reference hands to its Processor (:340-348, `build_cropping_network` above); the post-processing itself only uses
    the Processor's un-crop arithmetic, so None is allowed."""

    def __init__(self, plans, image_size, crop_size, device=None, cropping_network=None):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.processor = Processor(crop_size=crop_size, image_size=image_size, cropping_network=cropping_network)
        self.full_image_size = image_size
        if plans.get("transpose_forward") is None or plans.get("transpose_backward") is None:
            plans = dict(plans, transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2])                 # :346-352
        self.transpose_backward = list(plans["transpose_backward"])
        exp = plans.get("segmentation_export_params")
        if exp:                                                                                             # :354-361
            self.force_separate_z, self.interpolation_order, self.interpolation_order_z = (exp["force_separate_z"], exp["interpolation_order"],
                                                                                           exp["interpolation_order_z"])
        else:
            self.force_separate_z, self.interpolation_order, self.interpolation_order_z = None, 1, 0

    # -- per-file geometry chain shared by postprocess / postprocess_no_seg
    def _to_native(self, arr_chwd, padding_need, size_before):
        a = uncrop_stack(arr_chwd, padding_need, self.processor, self.device)
        assert list(a.shape[1:-1]) == [self.full_image_size, self.full_image_size]
        a = resize_with_pad_or_crop(a, size_before)
        assert list(size_before) == list(a.shape[1:])
        a = a.transpose((0, 3, 1, 2))                                                                       # C, depth, H, W
        return a.transpose([0] + [i + 1 for i in self.transpose_backward])

    @staticmethod
    def _load_pkl(pkl_path, fname):
        properties = load_plain_pickle(join(pkl_path, fname + ".pkl"))     # plain values only: anything else raises UnpicklingError
        return properties, np.asarray(properties["padding_need"]), list(properties["voxelmorph_size_before"])

    def postprocess(self, pred_path_list_registered, pred_path_list_seg, pred_path_list_flow, pred_path_list_seg_ed, pkl_path,
                    newpath_flow, newpath_registered, newpath_seg, patient_name):
        """voxelmorph_saver_Lib.py:119-277."""
        newpath_flow, newpath_registered, newpath_seg = (join(p, patient_name) for p in (newpath_flow, newpath_registered, newpath_seg))
        for p in (newpath_flow, newpath_registered, newpath_seg):
            delete_if_exist(p)
            os.makedirs(p)
        assert len(pred_path_list_registered) == len(pred_path_list_seg) == len(pred_path_list_flow)
        for seg_path in pred_path_list_seg_ed:                                                              # :145-186 (ED frame: segmentation only)
            fname = os.path.basename(seg_path)[:-4]
            properties, padding_need, size_before = self._load_pkl(pkl_path, fname)
            soft = self._to_native(np.load(seg_path)["seg"], padding_need, size_before)
            save_segmentation_nifti_from_softmax(soft, join(newpath_seg, fname + ".nii.gz"), properties, self.interpolation_order, None, None, None,
                                                 None, None, self.force_separate_z, self.interpolation_order_z, False, None, None, None, None)
        for registered_path, flow_path, seg_path in zip(pred_path_list_registered, pred_path_list_flow, pred_path_list_seg):
            fname = os.path.basename(seg_path)[:-4]
            properties, padding_need, size_before = self._load_pkl(pkl_path, fname)
            flow = self._to_native(np.load(flow_path)["flow"].transpose(3, 0, 1, 2), padding_need, size_before)       # [H,W,D,2] -> [2,H,W,D]
            reg = self._to_native(_nib_read(registered_path)[None], padding_need, size_before)
            soft = self._to_native(np.load(seg_path)["seg"], padding_need, size_before)
            save_segmentation_nifti_from_softmax(soft, join(newpath_seg, fname + ".nii.gz"), properties, self.interpolation_order, None, None, None,
                                                 None, None, self.force_separate_z, self.interpolation_order_z, False, flow,
                                                 join(newpath_flow, fname + ".npz"), reg, join(newpath_registered, fname + ".nii.gz"))
        determine_postprocessing_custom(newpath_registered, "", final_subf_name="_postprocessed", debug=True)
        determine_postprocessing_custom(newpath_seg, "", final_subf_name="_postprocessed", debug=True)

    def postprocess_no_seg(self, pred_path_list_registered, pred_path_list_flow, pkl_path, newpath_flow, newpath_registered, patient_name,
                           scratch_seg_dir=None):
        """voxelmorph_saver_Lib.py:20-115: flow and propagated labels only (the all-zero segmentation the reference still hands to the
        exporter goes to `scratch_seg_dir`, default a `_no_seg` folder beside the Registered one)."""
        newpath_flow, newpath_registered = join(newpath_flow, patient_name), join(newpath_registered, patient_name)
        scratch = scratch_seg_dir or join(os.path.dirname(os.path.dirname(newpath_registered)), "_no_seg", patient_name)
        for p in (newpath_flow, newpath_registered, scratch):
            delete_if_exist(p)
            os.makedirs(p)
        assert len(pred_path_list_registered) == len(pred_path_list_flow)
        for registered_path, flow_path in zip(pred_path_list_registered, pred_path_list_flow):
            fname = os.path.basename(registered_path)[:-7]
            properties, padding_need, size_before = self._load_pkl(pkl_path, fname)
            flow = self._to_native(np.load(flow_path)["flow"].transpose(3, 0, 1, 2), padding_need, size_before)
            reg = self._to_native(_nib_read(registered_path)[None], padding_need, size_before)
            save_segmentation_nifti_from_softmax(np.zeros_like(reg), join(scratch, fname + ".nii.gz"), properties, self.interpolation_order, None,
                                                 None, None, None, None, self.force_separate_z, self.interpolation_order_z, False, flow,
                                                 join(newpath_flow, fname + ".npz"), reg, join(newpath_registered, fname + ".nii.gz"))
        determine_postprocessing_custom(newpath_registered, "", final_subf_name="_postprocessed", debug=True)


def run(pred_path, pkl_path, plans, image_size, crop_size, no_seg=False, device=None, cropping_network=None):
    """voxelmorph_saver_Lib.py:284-394 (__main__) for the folder that holds `Raw/`: creates `Postprocessed/{Flow,Registered,Segmentation}`
    and processes every patient folder of `Raw/Registered`."""
    from glob import glob
    output_dir = join(pred_path, "Postprocessed")
    delete_if_exist(output_dir)
    newpath = {k: join(output_dir, k) for k in ("Flow", "Registered", "Segmentation")}
    for p in newpath.values():
        os.makedirs(p)
    saver = Saver(plans, image_size, crop_size, device, cropping_network)
    registered_dir = join(pred_path, "Raw", "Registered")
    patients = sorted(n for n in os.listdir(registered_dir) if os.path.isdir(join(registered_dir, n)))
    for patient_name in patients:
        regs = sorted(glob(join(pred_path, "Raw", "Registered", patient_name, "*.gz")))
        segs = sorted(glob(join(pred_path, "Raw", "Segmentation", patient_name, "*.npz")))
        flows = sorted(glob(join(pred_path, "Raw", "Flow", patient_name, "*.npz")))
        reg_names = [os.path.basename(x)[:-7] for x in regs]
        seg_ed = [x for x in segs if os.path.basename(x)[:-4] not in reg_names]                              # :376-378
        segs = [x for x in segs if os.path.basename(x)[:-4] in reg_names]
        if no_seg:
            saver.postprocess_no_seg(regs, flows, pkl_path, newpath["Flow"], newpath["Registered"], patient_name)
        else:
            saver.postprocess(regs, segs, flows, seg_ed, pkl_path, newpath["Flow"], newpath["Registered"], newpath["Segmentation"], patient_name)
    return output_dir


def main(argv=None):
    """CLI of voxelmorph_saver_Lib.py (--no_seg, --dataset) with the hard-coded paths of its __main__ as arguments."""
    import argparse
    import json
    parser = argparse.ArgumentParser()
    parser.add_argument("--no_seg", required=False, action="store_true", help="Whether to save direct segmentation")
    parser.add_argument("--dataset", required=True, help="Dataset name (Lib: image 384 / crop 192; ACDC: image 224 / crop 128)")
    parser.add_argument("--pred_path", required=True, help="folder that holds Raw/")
    parser.add_argument("--pkl_path", required=True, help="folder with the per-file .pkl properties")
    parser.add_argument("--plans", required=True, help="plans.json of the model folder")
    parser.add_argument("--image_size", type=int, default=None)
    parser.add_argument("--crop_size", type=int, default=None)
    parser.add_argument("--cropper_config", default=None, help="adversarial_acdc.yaml of the cropping network (the reference reads it from the cwd, :340)")
    parser.add_argument("--cropper_weights", default=None, help="<cropper_weights_folder_path>/model_final_checkpoint.model (:344)")
    parser.add_argument("--window_size", type=int, default=None)
    a = parser.parse_args(argv)
    sizes = {"Lib": (384, 192, 8), "ACDC": (224, 128, 7)}                                                    # :301-320
    image_size, crop_size, window_size = sizes.get(a.dataset, (None, None, None))
    image_size, crop_size, window_size = a.image_size or image_size, a.crop_size or crop_size, a.window_size or window_size
    assert image_size and crop_size, "unknown dataset: pass --image_size and --crop_size"
    with open(a.plans) as f:
        plans = json.load(f)
    net = build_cropping_network(a.cropper_config, image_size, window_size, a.cropper_weights) if a.cropper_config else None
    return run(a.pred_path, a.pkl_path, plans, image_size, crop_size, a.no_seg, cropping_network=net)


if __name__ == "__main__":
    main()
