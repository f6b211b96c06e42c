"""Predict-from-folder API of the reference, on the HIP path.

Mirrors `nnunet/inference/predict.py` (public names, argument lists and defaults of `predict_from_folder` :665-672,
`predict_cases` :228-232, `check_input_folder_and_return_caseIDs` :629, the CLI flags of :782-858) and the output
layout the downstream scripts read (`<out>/<patient>/{Segmentation,Registered}/<case>.nii.gz` uint8 and
`<out>/<patient>/Flow/<case>.npz` with `flow` float32 `[Y,X,Z,2]` + `spacing`, segmentation_export.py:190-219).

Differences that are deliberate and documented in DESIGN.md:
  * every patient folder is processed (the reference `return`s from inside its patient loop, predict.py:743);
  * the model folder holds `plans.json` + `fold_X/<chk>.model` written by `save_model_folder` below (a plain tensor
    dict, loaded with `torch.load(weights_only=True)`); the reference's `plans.pkl` / `*.model.pkl` are pickles of
    trainer objects whose classes cannot be imported here, and nothing on this path un-pickles;
  * preprocessing is the z-score of the volume at native spacing (crop-to-nonzero and resampling are the "next" rows
    of SURVEY.md section 8f), so the exporter needs no resampling either; the heart centroid of `Processor` comes from
    the image centre because the reference's 2-class cropping network is outside this path;
  * when no ED label map is supplied the ED segmentation predicted by the U-Net is the one propagated;
  * `predict_from_folder` loads the model once and fills the device batch ACROSS patients: the cropped slices of as many patients as fit
    `MAX_SLICES_PER_LAUNCH` (64) go through the networks as one batch (`CineTrainer.predict_patients_flow`), the next group's files are
    read and preprocessed by the `num_threads_preprocessing` pool meanwhile, and the NIfTI / NPZ export of finished patients runs in the
    `num_threads_nifti_save` pool while the device works on the next group.  The reference predicts and exports patient by patient
    (predict.py:228-354, :1008-1110); per-patient results are those of the one-patient call up to the launch shapes the batch size
    selects (tests/test_predict_api.py asserts the bound).  `LAST_TIMING` holds the wall-time split of the last call.
"""
import argparse
import csv
import glob
import json
import os
import shutil
from copy import deepcopy
from multiprocessing.pool import ThreadPool

import numpy as np
import torch

from . import ops
from .inference import Processor, chunk_orders, normalize_intensity_, pad_nd_image, predict_3D_2Dconv_tiled, predict_cine_slices
from .models import Generic_UNet, SegFlowGaussian
from .nifti import read_nifti, write_nifti

join = os.path.join

MAX_SLICES_PER_LAUNCH = int(os.environ.get("CF_API_SLICES", "64"))   # cropped cine slices per device batch of the file-level API
FIRST_BATCH_SLICES = int(os.environ.get("CF_API_FIRST_SLICES", "16"))     # slices of the first device batch of a predict_cases / predict_from_folder call
GIL_SWITCH_INTERVAL = float(os.environ.get("CF_API_SWITCH_INTERVAL", "0.0002"))   # seconds; Python's default is 0.005
_STREAM_POOL = {}                                                    # (device index) -> [torch.cuda.Stream]: reused by the preprocessing threads of every call
_STREAM_POOL_LOCK = None


def _pooled_stream(device, k):
    global _STREAM_POOL_LOCK
    import threading
    if _STREAM_POOL_LOCK is None:
        _STREAM_POOL_LOCK = threading.Lock()
    with _STREAM_POOL_LOCK:
        pool = _STREAM_POOL.setdefault(torch.device(device).index or 0, [])
        while len(pool) <= k:
            pool.append(torch.cuda.Stream(device=device))
        return pool[k]


# "batch" (default): the pool runs one device batch ahead of the collector; "all": the whole request is read and preprocessed before the first
# batch -- the networks then run at their device-only rate (5.6 instead of 8.1 s for 16 patients) but the reading is not hidden (4.9 s):
# 310 against 361 frames/s (profiles/r03_api_split.md)
PREFETCH_ALL = os.environ.get("CF_API_PREFETCH", "batch") == "all"
class _StackSampler:
    """CF_API_PROFILE=2: every 2 ms the innermost three frames of all other threads (sys._current_frames), as a histogram on stderr"""

    def __init__(self):
        import collections
        import threading
        self.hist, self.stop, self.me = collections.Counter(), False, threading.get_ident()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        import sys
        import threading
        import time
        own = threading.get_ident()
        while not self.stop:
            for tid, f in sys._current_frames().items():
                if tid in (own, self.me):
                    continue
                chain = []
                while f is not None and len(chain) < 3:
                    chain.append("%s:%d %s" % (os.path.basename(f.f_code.co_filename), f.f_lineno, f.f_code.co_name))
                    f = f.f_back
                self.hist[" <- ".join(chain)] += 1
            time.sleep(0.002)

    def report(self, title):
        import sys
        self.stop = True
        self.t.join()
        tot = sum(self.hist.values())
        print("== " + title, file=sys.stderr)
        for k, v in self.hist.most_common(16):
            print("   %5.1f %%  %s" % (100.0 * v / max(tot, 1), k[:220]), file=sys.stderr)


API_PROFILE = os.environ.get("CF_API_PROFILE", "0") != "0"
DEVICE_SPLIT = {}                                                    # CF_API_PROFILE=1: prepare / networks / finish seconds inside the device batches
LAST_TIMING = {}                                                     # wall-time split of the last predict_from_folder / predict_cases call


# ------------------------------------------------------------------------------------------------ model folder
def save_model_folder(folder, seg_net, flow_net, plans, fold=0, checkpoint_name="model_final_checkpoint", seg_sd=None, flow_sd=None, crop_sd=None):
    """Write `<folder>/plans.json` and `<folder>/fold_<fold>/<checkpoint_name>.model` (state dicts keyed by the
    reference's parameter names).  `seg_sd` / `flow_sd` / `crop_sd` (the Processor's cropping network, plans['cropping_net']): {name: tensor}."""
    os.makedirs(join(folder, "fold_%d" % fold), exist_ok=True)
    with open(join(folder, "plans.json"), "w") as f:
        json.dump(plans, f, indent=1)
    ck = {"seg_state_dict": {k: v.cpu() for k, v in seg_sd.items()}, "flow_state_dict": {k: v.cpu() for k, v in flow_sd.items()}}
    if crop_sd is not None:
        ck["crop_state_dict"] = {k: v.cpu() for k, v in crop_sd.items()}
    torch.save(ck, join(folder, "fold_%d" % fold, checkpoint_name + ".model"))


def default_plans(image_size=256, crop_size=None, flow_variant="video", seg_base=32, seg_pool=6, reduced=None):
    p = {"num_modalities": 1, "num_classes": 4, "patch_size": [image_size, image_size], "transpose_forward": [0, 1, 2],
         "transpose_backward": [0, 1, 2], "mirror_axes": [0, 1], "crop_size": crop_size or image_size, "image_size": image_size,
         "seg_net": {"base_num_features": seg_base, "num_pool": seg_pool},
         "flow_net": {"variant": flow_variant, "kwargs": reduced or {}}}
    return p


def _config_values(spec, model_folder, reader):
    """a config given inline (the YAML's mapping) or as a file name, absolute or relative to the model folder"""
    if isinstance(spec, dict):
        return spec
    path = spec if os.path.isabs(spec) or model_folder is None else join(model_folder, spec)
    return reader(path)


class ModelWrapFlow:
    """ModelWrap (successive.yaml: Optical_flow_model_successive.py:58-134) behind the flow-network interface of predict_cine_slices:
    __call__(x [T,B,1,H,W]) -> {'backward_flow': ED->t cumulative flow [T-1,B,2,H,W]} (out2['cumulated'], or model1's single pair flow
    when T == 2, :95-96)."""
    num_classes = 4

    def __init__(self, wrap):
        self.wrap = wrap

    def __call__(self, x):
        _out1, out2 = self.wrap(x, inference=False)
        return {"backward_flow": out2["cumulated"] if x.shape[0] > 2 else out2["flow"][None]}

    def state_shapes(self):
        return self.wrap.state_shapes()

    def load_state_dict(self, sd, device, **kw):
        self.wrap.load_state_dict(sd, device, **kw)
        return self


# `mixed_precision=True` (the reference's default) is honoured only on request: measured on the seeded networks, the one-term segmentation path
# misses the 1e-3 Dice bar (per-class Dice 0.992-0.999 against the f32-class path, tests/test_gpu_models.py::test_generic_unet_mixed_precision_measured,
# bench.py --seg-precision f16), so by default the flag is accepted and every network stays f32-class.  CF_SEG_MIXED_PRECISION=1 turns it on.
SEG_MIXED_PRECISION = os.environ.get("CF_SEG_MIXED_PRECISION", "0") == "1"


class CineTrainer:
    """Duck-types the trainer interface `predict_cases` uses (predict.py:285-354, :1028-1091).

    plans['flow_net'] selects the flow network either the build's short way, {'variant': 'video' | 'raft_config', 'kwargs': {...}}, or the
    reference's way, {'config': <mapping of the YAML's values, or a file name such as 'config.yaml' in the model folder>}: that config goes
    through cineflow.config (`read_config_video` + `build_seg_flow_gaussian_model` / the successive pair), as run_training.py:191 does with
    `<weights>/config.yaml`.  `prediction: false` is supplied when the file lacks it (raft_config.yaml).
    plans['cropping_net'] = {'type': 'mtl', 'config': <adversarial_acdc.yaml values or file name>, 'window_size': 7} puts the reference's own
    cropping network -- MTLmodel(num_classes=2), voxelmorph_saver_Lib.py:340-348 -- into the Processor; {'base_num_features', 'num_pool'}
    keeps the 2-class Generic_UNet stand-in of round 2."""

    def __init__(self, plans, device, model_folder=None):
        self.plans = plans
        self.device = device
        self.num_classes = plans["num_classes"]
        self.data_aug_params = {"mirror_axes": tuple(plans["mirror_axes"])}
        self.patch_size = tuple(plans["patch_size"])
        self.processor = Processor(crop_size=plans["crop_size"], image_size=plans["patch_size"][0])
        # mixed_precision of load_model_and_checkpoint_files / predict_from_folder (the reference's default True): with CF_SEG_MIXED_PRECISION=1
        # the segmentation U-Net runs its convolutions in the one-term fp16 product mode (ops.conv_terms(1)), like the reference's autocast on
        # that path (neural_network.py:140-146); the flow network never does (SegFlowGaussian.py:2905-2909).  Default: flag accepted, f32-class.
        self.mixed_precision = False
        self.crop_net = None
        ck = plans.get("cropping_net")
        if ck:
            from .inference import CroppingNet
            if ck.get("type") == "mtl":
                from . import config as C
                cfg = _config_values(ck["config"], model_folder, lambda f: C.read_config(f, False, False))
                self.crop_net = C.build_2d_model(cfg, conv_layer=None, norm=None, log_function=None, image_size=plans["patch_size"][0],
                                                 window_size=ck["window_size"], middle=False, num_classes=2, processor=None)
                self.processor.cropping_network = self.crop_net            # MTLmodel.forward returns {'pred': logits} itself
            else:
                self.crop_net = Generic_UNet(1, ck["base_num_features"], 2, ck["num_pool"])
                self.processor.cropping_network = CroppingNet(self.crop_net)
        sk = plans["seg_net"]
        self.seg_net = Generic_UNet(plans["num_modalities"], sk["base_num_features"], self.num_classes, sk["num_pool"],
                                    pool_op_kernel_sizes=sk.get("pool_op_kernel_sizes"))     # the plans' per-stage pooling (plans_per_stage[...]['pool_op_kernel_sizes'])
        fk = plans["flow_net"]
        if fk.get("config") is not None:
            from . import config as C
            cfg = C.with_defaults(_config_values(fk["config"], model_folder, C.read_config_video), prediction=False)
            net = C.build_flow_net(cfg, image_size=plans["crop_size"])
            self.flow_net = ModelWrapFlow(net) if not isinstance(net, SegFlowGaussian) else net
        else:
            ma = fk["variant"] == "raft_config"
            kw = dict(image_size=plans["crop_size"], motion_appearance=ma, dim_feedforward=3072 if ma else 2048)
            kw.update(fk.get("kwargs", {}))
            self.flow_net = SegFlowGaussian(**kw)

    # -- network_trainer.py:418 load_checkpoint_ram(params, train)
    def load_checkpoint_ram(self, params, train=False):
        self.seg_net.load_state_dict(params["seg_state_dict"], self.device)
        self.flow_net.load_state_dict(params["flow_state_dict"], self.device)
        if self.crop_net is not None:
            if "crop_state_dict" not in params:
                raise KeyError("plans['cropping_net'] is set but the checkpoint has no 'crop_state_dict' (save_model_folder(..., crop_sd=...))")
            self.crop_net.load_state_dict(params["crop_state_dict"], self.device)

    # -- nnUNetTrainer.py:571-597 preprocess_patient(list_of_files) -> (data[C,Z,Y,X], seg, properties)
    def preprocess_patient(self, input_files):
        """Crop to non-zero, resample to the stage's spacing and normalise on the device (cineflow.preprocessing), driven by the
        same plan entries as the reference: preprocessor_name (default PreprocessorFor2D -- the fork's networks are 2-D),
        normalization_schemes, use_mask_for_norm, transpose_forward, dataset_properties.intensityproperties and
        plans_per_stage[stage].current_spacing (absent: the case keeps its own spacing)."""
        from . import preprocessing as P
        plans = self.plans
        nmod = plans["num_modalities"]
        as_int_keys = lambda d, default: {int(k): v for k, v in (d or {c: default for c in range(nmod)}).items()}   # noqa: E731  (JSON keys are strings)
        schemes = as_int_keys(plans.get("normalization_schemes"), "nonCT")
        use_mask = as_int_keys(plans.get("use_mask_for_norm"), False)
        ip = (plans.get("dataset_properties") or {}).get("intensityproperties")
        ip = None if ip is None else {int(k): v for k, v in ip.items()}
        name = plans.get("preprocessor_name") or "PreprocessorFor2D"
        cls = getattr(P, name, None)
        assert cls is not None, "Could not find preprocessor %s in cineflow.preprocessing" % name
        pre = cls(schemes, use_mask, list(plans["transpose_forward"]), ip)
        stages = plans.get("plans_per_stage")
        if stages:
            st = stages[str(plans.get("stage", 0))] if isinstance(stages, dict) and str(plans.get("stage", 0)) in stages else stages[plans.get("stage", 0)]
            return pre.preprocess_test_case(list(input_files), np.array(st["current_spacing"], dtype=float))
        return pre.preprocess_test_case(list(input_files), None)      # (no stages in the plans: the case keeps its own spacing)

    # -- nnUNetTrainer.py:637-679
    def predict_preprocessed_data_return_seg_and_softmax(self, data, do_mirroring=True, mirror_axes=None, use_sliding_window=True,
                                                         step_size=0.5, use_gaussian=True, pad_border_mode="constant", pad_kwargs=None,
                                                         all_in_gpu=False, verbose=True, mixed_precision=True):
        mirror_axes = self.data_aug_params["mirror_axes"] if mirror_axes is None else mirror_axes
        with ops.conv_terms(1 if (mixed_precision and SEG_MIXED_PRECISION) else 3):
            return predict_3D_2Dconv_tiled(self.seg_net, data, self.patch_size, step_size=step_size, do_mirroring=do_mirroring,
                                           mirror_axes=mirror_axes, use_gaussian=use_gaussian, pad_border_mode=pad_border_mode,
                                           pad_kwargs=pad_kwargs)

    # -- SegFlowGaussian.py:3294-3533 up to the network call: pad, centre crop to the patch, heart-centred crop, z-score
    def _flow_prepare(self, unlabeled, target, processor, pad_border_mode, pad_kwargs, centroid):
        T, _, Z, Y, X = unlabeled.shape
        P = self.patch_size
        x = unlabeled[:, 0]                                                            # [T,Z,Y,X]
        data, slicer = pad_nd_image(x, P, pad_border_mode, pad_kwargs, True)            # SegFlowGaussian.py:3310
        Hp, Wp = data.shape[-2:]
        y1, y2 = int(Hp / 2 - P[0] / 2), int(Hp / 2 + P[0] / 2)                         # :3391-3397 centre crop to the patch
        x1, x2 = int(Wp / 2 - P[1] / 2), int(Wp / 2 + P[1] / 2)
        dev = self.device
        patch = torch.from_numpy(np.ascontiguousarray(data[:, :, y1:y2, x1:x2])).to(dev, dtype=torch.float32)   # [T,Z,P,P]
        cs = processor.crop_size
        # one cropping window per slice (SegFlowGaussian.py:3099-3103 runs per slice): around the caller's centroid, else around the mean
        # centroid of the cropping network's masks (processor.py:232-237), else around the patch centre
        wins = []
        for z in range(Z):
            if centroid is not None:
                cen = centroid
            elif getattr(processor, "cropping_network", None) is not None:
                cen = [int(v) for v in processor.preprocess_no_registration(patch[:, z].unsqueeze(1).contiguous())[0]]
            else:
                cen = (P[1] // 2, P[0] // 2)
            wins.append(processor.adjust_cropping_window(cen))
        crop = torch.empty((T, Z, cs, cs), dtype=torch.float32, device=dev)
        for z in range(Z):
            cx0, cx1, cy0, cy1 = wins[z]["crop_indices"]
            blk = ops.crop2d(patch[:, z].contiguous(), cy0, cx0, cs, cs)               # [T,cs,cs]
            normalize_intensity_(blk)                                                   # :3108 NormalizeIntensity on the slice's [T,h,w] block
            crop[:, z] = blk
        ed = None
        if target is not None:
            tp = pad_nd_image(np.asarray(target)[None], P, "constant", {"constant_values": 0}, False)[0]
            tp = tp[:, y1:y2, x1:x2]
            ed = torch.from_numpy(np.ascontiguousarray(np.stack([tp[z, wins[z]["crop_indices"][2]:wins[z]["crop_indices"][3],
                                                                    wins[z]["crop_indices"][0]:wins[z]["crop_indices"][1]] for z in range(Z)]))).to(dev, dtype=torch.uint8)
        pad_need = np.stack([np.asarray(w["padding_need"], dtype=np.int64) for w in wins], axis=1)     # [4, Z]
        return {"frames": crop.view(T, Z, 1, cs, cs), "ed": ed, "pad_need": pad_need, "slicer": slicer, "geom": (T, Z, Y, X, Hp, Wp, y1, y2, x1, x2),
                "processor": processor}

    @staticmethod
    def _to_host(tensors):
        """device tensors -> numpy arrays through pinned staging buffers (torch's caching host allocator), one synchronisation for all of
        them: the per-patient results are ~0.5 GB, pageable `.cpu()` copies were a fifth of the API's device stage"""
        hosts = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            hosts.append(h)
        torch.cuda.current_stream().synchronize()
        return [h.numpy() for h in hosts]

    # -- :3427-3467 after the network call: per-slice uncrop, centre window, un-pad, host copies
    def _flow_finish(self, prep, out, return_crop, want_raw=True, want_softmax=True):
        T, Z, Y, X, Hp, Wp, y1, y2, x1, x2 = prep["geom"]
        processor, pad_need, slicer, frames, dev = prep["processor"], prep["pad_need"], prep["slicer"], prep["frames"], self.device

        def place(t):  # [T, C?, Z, cs, cs] -> [..., Z, Y, X]: per-slice uncrop (processor.py:178-186), centre window, un-pad
            zax = t.dim() - 3
            full = torch.stack([processor.uncrop_no_registration(t.select(zax, z).contiguous(), pad_need[:, z]) for z in range(Z)], dim=zax)
            canvas = torch.zeros(tuple(full.shape[:-2]) + (Hp, Wp), dtype=full.dtype, device=dev)
            canvas[..., y1:y2, x1:x2] = full
            return canvas[..., slicer[-2], slicer[-1]]

        softmax = place(out["softmax"].permute(0, 2, 1, 3, 4).contiguous())            # [T,K,Z,Y,X]
        flow = place(out["flow"].permute(0, 2, 1, 3, 4).contiguous())                  # [T,2,Z,Y,X]
        reg = place(out["registered"].float())[:, None]                                # [T,1,Z,Y,X]
        seg = ops.argmax_channels(softmax.reshape(T, self.num_classes, -1).contiguous()).view(T, Z, Y, X)
        # (want_softmax=False: the exporter will write the device arg-max `seg`; the [T,K,Z,Y,X] probabilities -- 200 MB per patient -- stay on the device)
        dev_out = [seg, softmax.contiguous() if want_softmax else torch.empty(0, device=dev), flow.contiguous(), reg.contiguous()]
        if want_raw:
            dev_out.append(torch.cat([frames.permute(0, 2, 1, 3, 4), out["flow"].permute(0, 2, 1, 3, 4)], 1))
        if return_crop:
            dev_out += [out["softmax"].permute(0, 2, 1, 3, 4).contiguous(), out["flow"].permute(0, 2, 1, 3, 4).contiguous(), out["registered"].contiguous()]
        host = self._to_host(dev_out)
        if not want_softmax:
            host[1] = None
        res = tuple(host[:4]) + ((host[4],) if want_raw else (None,))
        if return_crop:
            c = host[-3:]
            return res + ({"softmax": c[0], "flow": c[1], "registered": c[2], "padding_need": pad_need, "size_before": [int(Y), int(X), int(Z)]},)
        return res

    # -- nnUNetTrainer.py:682-726 -> SegFlowGaussian.predict_3D_flow :2837, _internal_predict_2D_2Dconv_tiled_flow :3294-3533
    def predict_preprocessed_data_return_seg_and_softmax_flow(self, unlabeled, target=None, target_mask=None, processor=None,
                                                              do_mirroring=True, mirror_axes=None, use_sliding_window=True, step_size=0.5,
                                                              use_gaussian=True, pad_border_mode="constant", pad_kwargs=None,
                                                              all_in_gpu=False, verbose=True, mixed_precision=True, centroid=None, return_crop=False):
        """unlabeled [T,1,Z,Y,X] (numpy) -> (seg [T,Z,Y,X], softmax [T,K,Z,Y,X], flow [T,2,Z,Y,X], registered [T,1,Z,Y,X],
        raw [T,3,Z,crop,crop]).  target: optional ED label volume [Z,Y,X].  centroid: (x, y) of the heart in the patch, or None (patch
        centre).  return_crop=True appends the crop-space results the voxelmorph_saver layout stores: dict(softmax [T,K,Z,c,c],
        flow [T,2,Z,c,c], registered [T,Z,c,c], padding_need [4,Z], size_before [Y,X,Z])."""
        return self.predict_patients_flow([unlabeled], [target], processor=processor, do_mirroring=do_mirroring, mirror_axes=mirror_axes,
                                          pad_border_mode=pad_border_mode, pad_kwargs=pad_kwargs, centroids=[centroid], return_crop=return_crop)[0]

    def predict_patients_flow(self, unlabeled_list, targets=None, processor=None, do_mirroring=True, mirror_axes=None, pad_border_mode="constant",
                              pad_kwargs=None, centroids=None, return_crop=False, want_raw=True, want_softmax=True):
        """The one-patient call above for several patients whose cropped slices share ONE device batch: every patient is padded / cropped /
        z-scored on its own (`_flow_prepare`), the `[T, Z_p, 1, c, c]` stacks of the patients with the same frame count T are concatenated
        on the slice axis, predict_cine_slices runs once per such group, and each patient's slices go back through its own un-crop
        (`_flow_finish`).  No kernel mixes batch entries; results are those of the one-patient calls up to the launch shapes the batch
        size selects.  Returns one result tuple per patient, in order."""
        processor = processor or self.processor
        mirror_axes = self.data_aug_params["mirror_axes"] if mirror_axes is None else mirror_axes
        n = len(unlabeled_list)
        targets = targets or [None] * n
        centroids = centroids or [None] * n
        import time
        prof = API_PROFILE                                  # CF_API_PROFILE=1: synchronise between the stages and add their times to DEVICE_SPLIT
        t0 = time.perf_counter()
        preps = [self._flow_prepare(u, t, processor, pad_border_mode, pad_kwargs, c) for u, t, c in zip(unlabeled_list, targets, centroids)]
        if prof:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            DEVICE_SPLIT["prepare_s"] = DEVICE_SPLIT.get("prepare_s", 0.0) + t1 - t0
        outs = [None] * n
        by_T = {}
        for i, pr in enumerate(preps):
            by_T.setdefault((pr["frames"].shape[0], pr["ed"] is not None), []).append(i)
        for (_T, has_ed), idx in by_T.items():
            frames = preps[idx[0]]["frames"] if len(idx) == 1 else torch.cat([preps[i]["frames"] for i in idx], dim=1)
            ed = None if not has_ed else (preps[idx[0]]["ed"] if len(idx) == 1 else torch.cat([preps[i]["ed"] for i in idx], dim=0))
            out = predict_cine_slices(self.flow_net, self.seg_net, frames.contiguous(), ed, do_mirroring, mirror_axes,
                                      seg_mixed_precision=bool(self.mixed_precision and SEG_MIXED_PRECISION))
            z0 = 0
            for i in idx:
                Z = preps[i]["frames"].shape[1]
                outs[i] = {k: v[:, z0:z0 + Z] for k, v in out.items()}
                z0 += Z
        if prof:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            DEVICE_SPLIT["networks_s"] = DEVICE_SPLIT.get("networks_s", 0.0) + t2 - t1
        res = [self._flow_finish(pr, o, return_crop, want_raw, want_softmax) for pr, o in zip(preps, outs)]
        if prof:
            torch.cuda.synchronize()
            DEVICE_SPLIT["finish_s"] = DEVICE_SPLIT.get("finish_s", 0.0) + time.perf_counter() - t2
        return res


def _fold_dirs(folder, folds):
    if folds is None or folds == "None":
        return sorted(d for d in os.listdir(folder) if d.startswith("fold_"))
    if isinstance(folds, (list, tuple)):
        return ["fold_%s" % i if str(i) != "all" else "all" for i in folds]
    return ["fold_%s" % folds]


_CHECKPOINT_PARTS = (("seg_state_dict", "seg_net"), ("flow_state_dict", "flow_net"), ("crop_state_dict", "crop_net"))


def _broadcast_params(trainer, params, nfolds, device):
    """The one collective of the multi-GPU path (the reference has no hook: every `--part_id` process of
    predict.py:806-821 reads the checkpoint itself): rank 0 holds `params` (the folds' checkpoint dicts), every rank gets each fold's weights
    as ONE flat fp32 broadcast (RCCL over xGMI under the nccl backend, gloo on CPU tensors) and rebuilds the dicts `load_checkpoint_ram` takes.
    Shapes come from the networks every rank built from plans.json, so ranks >= 1 need no checkpoint file."""
    from . import parallel
    shapes = {}
    for key, attr in _CHECKPOINT_PARTS:
        net = getattr(trainer, attr, None)
        if net is not None:
            for k, v in net.state_shapes().items():
                if not k.endswith("grid"):                                       # SpatialTransformer grids are rebuilt, never loaded
                    shapes[key + "/" + k] = v
    import torch.distributed as dist
    rank = dist.get_rank()
    out = []
    for f in range(nfolds):
        flat = None
        if rank == 0:
            flat = {}
            for key, _ in _CHECKPOINT_PARTS:
                for k, v in (params[f].get(key) or {}).items():
                    if key + "/" + k in shapes:
                        flat[key + "/" + k] = v
            missing = sorted(set(shapes) - set(flat))
            if missing:
                raise KeyError("checkpoint lacks %d tensors the networks of plans.json need, e.g. %s" % (len(missing), missing[:3]))
        got = parallel.broadcast_state_dict(flat, shapes, device)
        p = {}
        for name, t in got.items():
            key, k = name.split("/", 1)
            p.setdefault(key, {})[k] = t
        out.append(p)
    return out


def load_model_and_checkpoint_files(folder, folds=None, mixed_precision=None, checkpoint_name="model_final_checkpoint", device=None):
    """model_restore.py:109-155 equivalent for the plans.json / *.model folder format -> (trainer, [params per fold]).
    In a multi-process job (WORLD_SIZE > 1, one process per GPU) only rank 0 reads `fold_X/<checkpoint_name>.model`; the other ranks need
    plans.json alone and receive the weights through cineflow.parallel.broadcast_state_dict before the patient loop."""
    assert os.path.isfile(join(folder, "plans.json")), "Folder with saved model weights must contain a plans.json file"
    with open(join(folder, "plans.json")) as f:
        plans = json.load(f)
    device = device or torch.device("cuda", torch.cuda.current_device())
    trainer = CineTrainer(plans, device, model_folder=folder)
    trainer.mixed_precision = bool(mixed_precision)
    from . import parallel
    rank, world, _ = parallel.init_from_env()
    if world > 1:
        import torch.distributed as dist
        fold_dirs = _fold_dirs(folder, folds) if rank == 0 else None
        n = torch.tensor([len(fold_dirs) if rank == 0 else 0], dtype=torch.int64, device=device if device.type == "cuda" else "cpu")
        dist.broadcast(n, src=0)
        params = ([torch.load(join(folder, f, checkpoint_name + ".model"), map_location="cpu", weights_only=True) for f in fold_dirs]
                  if rank == 0 else None)
        return trainer, _broadcast_params(trainer, params, int(n.item()), device)
    params = [torch.load(join(folder, f, checkpoint_name + ".model"), map_location="cpu", weights_only=True) for f in _fold_dirs(folder, folds)]
    return trainer, params


# ------------------------------------------------------------------------------------------------ export
RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3   # nnunet/configuration.py


def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """preprocessing.py:30-32."""
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    """preprocessing.py:35-37."""
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def save_segmentation_nifti_from_softmax(segmentation_softmax, out_fname, properties_dict, order=1, region_class_order=None,
                                         seg_postprogess_fn=None, seg_postprocess_args=None, resampled_npz_fname=None,
                                         non_postprocessed_fname=None, force_separate_z=None, interpolation_order_z=0, verbose=True,
                                         flow=None, flow_path=None, registered=None, registered_path=None, seg_precomputed=None):
    """segmentation_export.py:29-223: resample softmax / flow / registered labels back to the size before resampling (device
    kernels, cineflow.ops.resample_data_or_seg), rescale the flow to the new pixel grid, argmax, place into the crop bounding
    box, write uint8 NIfTI with the case's geometry; flow [2,Z,Y,X] -> npz `flow` [Y,X,Z,2] float32 + `spacing`."""
    if isinstance(segmentation_softmax, str):
        assert os.path.isfile(segmentation_softmax), "If isinstance(segmentation_softmax, str) then isfile(segmentation_softmax) must be True"
        del_file = segmentation_softmax
        segmentation_softmax = np.load(segmentation_softmax)
        os.remove(del_file)
    shape_after_crop = tuple(properties_dict.get("size_after_cropping"))
    if seg_precomputed is not None:
        # the caller already holds arg-max(softmax) at the size after cropping (computed on the device, first maximum like numpy) and needs
        # neither the resampling branch nor the npz: the probabilities never left the device
        assert segmentation_softmax is None and resampled_npz_fname is None and region_class_order is None
        assert tuple(seg_precomputed.shape) == shape_after_crop, "seg_precomputed must have the size after cropping"
        current_shape = (0,) + tuple(seg_precomputed.shape)
    else:
        current_shape = segmentation_softmax.shape
    shape_before_crop = properties_dict.get("original_size_of_raw_data")
    if any(i != j for i, j in zip(current_shape[1:], shape_after_crop)):
        if force_separate_z is None:                                             # segmentation_export.py:88-98
            if get_do_separate_z(properties_dict.get("original_spacing")):
                do_separate_z, lowres_axis = True, get_lowres_axis(properties_dict.get("original_spacing"))
            elif get_do_separate_z(properties_dict.get("spacing_after_resampling")):
                do_separate_z, lowres_axis = True, get_lowres_axis(properties_dict.get("spacing_after_resampling"))
            else:
                do_separate_z, lowres_axis = False, None
        else:
            do_separate_z = force_separate_z
            lowres_axis = get_lowres_axis(properties_dict.get("original_spacing")) if do_separate_z else None
        if lowres_axis is not None and len(lowres_axis) != 1:
            do_separate_z = False
        if verbose:
            print("separate z:", do_separate_z, "lowres axis", lowres_axis)
        seg_old_spacing = ops.resample_data_or_seg(segmentation_softmax, shape_after_crop, is_seg=False, axis=lowres_axis, order=order,
                                                   do_separate_z=do_separate_z, order_z=interpolation_order_z)
        if flow is not None:
            rescale_y = shape_after_crop[1] / flow.shape[2]
            rescale_x = shape_after_crop[2] / flow.shape[3]
            flow = ops.resample_data_or_seg(np.asarray(flow, np.float32), shape_after_crop, is_seg=False, axis=lowres_axis, order=order,
                                            do_separate_z=do_separate_z, order_z=interpolation_order_z)
            flow[0] = flow[0] * rescale_y                                        # segmentation_export.py:123-124
            flow[1] = flow[1] * rescale_x
        if registered is not None:
            registered = ops.resample_data_or_seg(np.asarray(registered), shape_after_crop, is_seg=True, axis=lowres_axis, order=0,
                                                  do_separate_z=do_separate_z, order_z=0)
    else:
        if verbose:
            print("no resampling necessary")
        seg_old_spacing = segmentation_softmax
    if resampled_npz_fname is not None:
        np.savez_compressed(resampled_npz_fname, softmax=seg_old_spacing.astype(np.float16))
    if seg_precomputed is not None:
        seg = np.asarray(seg_precomputed)
    elif region_class_order is None:
        seg = seg_old_spacing.argmax(0)
    else:
        seg = np.zeros(seg_old_spacing.shape[1:])
        for i, c in enumerate(region_class_order):
            seg[seg_old_spacing[i] > 0.5] = c
    bbox = properties_dict.get("crop_bbox")
    if bbox is not None:                                                         # segmentation_export.py:153-177
        bbox = [list(b) for b in bbox]
        for c in range(3):
            bbox[c][1] = int(np.min((bbox[c][0] + seg.shape[c], shape_before_crop[c])))
        sl = tuple(slice(b[0], b[1]) for b in bbox)
        full = np.zeros(shape_before_crop, dtype=np.uint8)
        full[sl] = seg
        seg = full
        if flow is not None:
            f_full = np.zeros([2] + list(shape_before_crop), dtype=np.float32)
            f_full[(slice(None),) + sl] = flow
            flow = f_full
        if registered is not None:
            r_full = np.zeros(shape_before_crop, dtype=np.uint8)
            r_full[sl] = registered[0]
            registered = r_full[None]
    if seg_postprogess_fn is not None:
        seg = seg_postprogess_fn(np.copy(seg), *seg_postprocess_args)
    geo = (properties_dict["itk_spacing"], properties_dict["itk_origin"], properties_dict["itk_direction"])
    write_nifti(out_fname, seg.astype(np.uint8), *geo)
    if flow is not None:
        np.savez(flow_path, flow=np.asarray(flow, np.float32).transpose(2, 3, 1, 0), spacing=properties_dict["itk_spacing"])
    if registered is not None:
        write_nifti(registered_path, np.asarray(registered[0]).astype(np.uint8), *geo)


# ------------------------------------------------------------------------------------------------ post-processing
def load_postprocessing(json_file):
    """connected_components.py:109-120."""
    import ast
    with open(json_file) as f:
        a = json.load(f)
    mv = ast.literal_eval(a["min_valid_object_sizes"]) if "min_valid_object_sizes" in a else None
    return a["for_which_classes"], mv


def load_remove_save(input_file, output_file, for_which_classes, minimum_valid_object_size=None):
    """connected_components.py:31-48: keep the largest connected component of each class (device kernels, cineflow.ops)."""
    img, props = read_nifti(input_file)
    volume_per_voxel = float(np.prod(props["itk_spacing"], dtype=np.float64))
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.from_numpy(np.ascontiguousarray(img.astype(np.uint8))).to(dev)
    fw = [tuple(c) if isinstance(c, list) else c for c in for_which_classes] if for_which_classes is not None else None
    mv = None
    if minimum_valid_object_size is not None:
        mv = {(tuple(k) if isinstance(k, list) else k): v for k, v in minimum_valid_object_size.items()}
    t, largest_removed, kept_size = ops.remove_all_but_the_largest_connected_component(t, fw, volume_per_voxel, mv)
    write_nifti(output_file, t.cpu().numpy(), props["itk_spacing"], props["itk_origin"], props["itk_direction"])
    return largest_removed, kept_size


# ------------------------------------------------------------------------------------------------ predict API
def subfiles(folder, suffix=None, join_=True, sort=True):
    res = [f for f in os.listdir(folder) if os.path.isfile(join(folder, f)) and (suffix is None or f.endswith(suffix))]
    if sort:
        res.sort()
    return [join(folder, f) for f in res] if join_ else res


def check_input_folder_and_return_caseIDs(input_folder, expected_num_modalities):
    """predict.py:629-662, message for message."""
    print("This model expects %d input modalities for each image" % expected_num_modalities)
    files = subfiles(input_folder, suffix=".nii.gz", join_=False, sort=True)
    maybe_case_ids = np.unique([i[:-12] for i in files])
    remaining = deepcopy(files)
    missing = []
    assert len(files) > 0, "input folder did not contain any images (expected to find .nii.gz file endings)"
    for c in maybe_case_ids:
        for n in range(expected_num_modalities):
            expected_output_file = c + "_%04.0d.nii.gz" % n
            if not os.path.isfile(join(input_folder, expected_output_file)):
                missing.append(expected_output_file)
            else:
                remaining.remove(expected_output_file)
    print("Found %d unique case ids, here are some examples:" % len(maybe_case_ids),
          np.random.choice(maybe_case_ids, min(len(maybe_case_ids), 10)))
    print("If they don't look right, make sure to double check your filenames. They must end with _0000.nii.gz etc")
    if len(remaining) > 0:
        print("found %d unexpected remaining files in the folder. Here are some examples:" % len(remaining),
              np.random.choice(remaining, min(len(remaining), 10)))
    if len(missing) > 0:
        print("Some files are missing:")
        print(missing)
        raise RuntimeError("missing files in input_folder")
    return maybe_case_ids


def _subfolder_path(path, sub):
    """predict.py:1059-1068 inserts the sub-folder as path component 2 of a relative path; this is the same place for
    `<out>/<patient>/<case>` and also works for absolute / nested output folders."""
    return join(os.path.dirname(path), sub, os.path.basename(path))


def get_ed_es_indices(csv_filepath):
    """predict.py:1196-1198: first row of the patient's csv, columns `ed_index`, `es_index`."""
    with open(csv_filepath) as f:
        rows = list(csv.DictReader(f))
    return int(float(rows[0]["ed_index"])), int(float(rows[0]["es_index"]))


def put_ed_first(current_list_of_lists, current_output_files, csv_filepath):
    """predict.py:1165-1193: rotate the frame list so that the end-diastolic frame comes first."""
    ed_index, _es = get_ed_es_indices(csv_filepath)
    order = list(range(ed_index, len(current_list_of_lists))) + list(range(0, ed_index))
    return [list(current_list_of_lists[i]) for i in order], [current_output_files[i] for i in order]


_VOXELMORPH_RAW = None


def set_voxelmorph_raw(pred_path, pkl_path=None):
    """Switch on (pred_path given) or off (None) the additional `<pred_path>/Raw/{Registered,Segmentation,Flow}/<patient>/` +
    `<pkl_path>/<case>.pkl` output of predict_flow -- the crop-space layout voxelmorph_saver_* post-processes.  A module-level switch, so
    that predict_from_folder / predict_cases keep the reference's argument lists; pkl_path defaults to <pred_path>/pkl."""
    global _VOXELMORPH_RAW
    _VOXELMORPH_RAW = None if pred_path is None else (pred_path, pkl_path or join(pred_path, "pkl"))


def _export_flow_patient(result, trainer, output_filenames, property_list, interpolation_order, force_separate_z, interpolation_order_z,
                         save_npz, pool):
    """predict.py:1084-1110 for one patient's device results: transpose back, submit one export job per frame to the pool.
    Returns (seg_paths, flow_paths, reg_paths, jobs)."""
    voxelmorph_raw = _VOXELMORPH_RAW
    seg, softmax, flow, registered, _raw = result[:5]
    crop_out = result[5] if len(result) > 5 else None
    have_softmax = softmax is not None          # None: no resampling and no npz asked for -> the frames are written from the device arg-max `seg`
    assert len(seg) == len(flow) == len(registered) and (not have_softmax or len(softmax) == len(flow))
    if voxelmorph_raw is not None:
        assert crop_out is not None, "set_voxelmorph_raw was switched on after the device stage of this patient"
        from .voxelmorph_saver import write_raw
        patient = os.path.basename(os.path.dirname(os.path.abspath(output_filenames[0])))
        write_raw(voxelmorph_raw[0], voxelmorph_raw[1], patient, [os.path.basename(o)[:-7] for o in output_filenames], crop_out["softmax"],
                  crop_out["flow"], crop_out["registered"], property_list, crop_out["padding_need"], crop_out["size_before"], ed_position=0)
    # back to the axis order of the files (predict.py:1084-1089): preprocessing applied plans['transpose_forward']
    if trainer.plans.get("transpose_forward") is not None:
        tb = [0] + [i + 1 for i in trainer.plans.get("transpose_backward")]
        if have_softmax:
            softmax = [np.ascontiguousarray(x.transpose(tb)) for x in softmax]
        else:
            seg = [np.ascontiguousarray(x.transpose(trainer.plans.get("transpose_backward"))) for x in seg]
        flow = [np.ascontiguousarray(x.transpose(tb)) for x in flow]
        registered = [np.ascontiguousarray(x.transpose(tb)) for x in registered]
    seg_paths, flow_paths, reg_paths, jobs = [], [], [], []
    for t in range(len(flow)):
        seg_path, flow_path, reg_path = (_subfolder_path(output_filenames[t], s_) for s_ in ("Segmentation", "Flow", "Registered"))
        seg_paths.append(seg_path)
        flow_paths.append(flow_path[:-7] + ".npz")
        reg_paths.append(reg_path)
        npz = seg_path[:-7] + ".npz" if save_npz else None
        jobs.append(pool.apply_async(_timed_export, (trainer.device, softmax[t] if have_softmax else None, seg_path, property_list[t], interpolation_order, None,
                                                     None, None, npz, None, force_separate_z, interpolation_order_z, False, flow[t], flow_paths[-1],
                                                     registered[t], reg_path, None if have_softmax else seg[t])))
    return seg_paths, flow_paths, reg_paths, jobs


def _timed_export(dev, *a):
    import time
    t0 = time.perf_counter()
    torch.cuda.set_device(dev)                                                   # per-thread state: the resampling kernels must run on the caller's GPU
    save_segmentation_nifti_from_softmax(*a)
    return time.perf_counter() - t0


def _finish_flow_patient(seg_paths, reg_paths, jobs, output_filenames, disable_postprocessing, model):
    """wait for a patient's export jobs, then predict.py:1139-1156 (largest-component filter when the model folder has a postprocessing.json)"""
    work = sum(j.get() for j in jobs)
    if not disable_postprocessing:
        pp_file = join(model, "postprocessing.json")
        if os.path.isfile(pp_file):
            print("postprocessing...")
            shutil.copy(pp_file, os.path.abspath(os.path.dirname(output_filenames[0])))
            for_which_classes, min_valid_obj_size = load_postprocessing(pp_file)
            for pth in seg_paths + reg_paths:
                load_remove_save(pth, pth, for_which_classes, min_valid_obj_size)
        else:
            print("WARNING! Cannot run postprocessing because the postprocessing file is missing (%s)" % model)
    return work


def predict_flow(d, trainer, output_filenames, property_list, do_tta, mixed_precision, params, interpolation_order, force_separate_z,
                 interpolation_order_z, all_in_gpu, step_size, save_npz, disable_postprocessing, model, pool):
    """predict.py:1008-1162 for one patient: `d[t]` = preprocessed frame t (ED first), all frames form the cine sequence.
    Writes <patient>/{Segmentation,Flow,Registered}/<case>; returns the three path lists.
    After set_voxelmorph_raw(pred_path, pkl_path) the crop-space predictions are additionally written as `<pred_path>/Raw/...` +
    `<pkl_path>/<case>.pkl`, the input layout of voxelmorph_saver_* (cineflow.voxelmorph_saver)."""
    unlabeled = np.stack(d) + 1e-8                                               # predict.py:1025
    print("predicting", output_filenames)
    result = trainer.predict_preprocessed_data_return_seg_and_softmax_flow(
        unlabeled=unlabeled, target=None, target_mask=None, processor=trainer.processor, do_mirroring=do_tta,
        mirror_axes=trainer.data_aug_params["mirror_axes"], use_sliding_window=True, step_size=step_size, use_gaussian=True,
        all_in_gpu=all_in_gpu, mixed_precision=mixed_precision, verbose=False, return_crop=True)
    seg_paths, flow_paths, reg_paths, jobs = _export_flow_patient(result, trainer, output_filenames, property_list, interpolation_order,
                                                                  force_separate_z, interpolation_order_z, save_npz, pool)
    print("inference done. Now waiting for the segmentation export to finish...")
    _finish_flow_patient(seg_paths, reg_paths, jobs, output_filenames, disable_postprocessing, model)
    return seg_paths, flow_paths, reg_paths


def predict_non_flow(d, trainer, output_filenames, property_list, do_tta, mixed_precision, params, interpolation_order, force_separate_z,
                     interpolation_order_z, all_in_gpu, step_size, save_npz, disable_postprocessing, model, pool):
    """predict.py:926-1005: segmentation only, frame by frame, sliding window + TTA, fold ensembling."""
    jobs = []
    for t, input_img in enumerate(d):
        print("predicting", output_filenames[t])
        trainer.load_checkpoint_ram(params[0], False)
        softmax = trainer.predict_preprocessed_data_return_seg_and_softmax(
            input_img, do_mirroring=do_tta, mirror_axes=trainer.data_aug_params["mirror_axes"], use_sliding_window=True,
            step_size=step_size, use_gaussian=True, all_in_gpu=all_in_gpu, mixed_precision=mixed_precision)[1]
        for p_ in params[1:]:
            trainer.load_checkpoint_ram(p_, False)
            softmax += trainer.predict_preprocessed_data_return_seg_and_softmax(
                input_img, do_mirroring=do_tta, mirror_axes=trainer.data_aug_params["mirror_axes"], use_sliding_window=True,
                step_size=step_size, use_gaussian=True, all_in_gpu=all_in_gpu, mixed_precision=mixed_precision)[1]
        if len(params) > 1:
            softmax /= len(params)
        npz = output_filenames[t][:-7] + ".npz" if save_npz else None
        jobs.append(pool.apply_async(save_segmentation_nifti_from_softmax,
                                     (softmax, output_filenames[t], property_list[t], interpolation_order, None, None, None, npz, None,
                                      force_separate_z, interpolation_order_z)))
    return jobs


_MODEL_CACHE = {}


def clear_model_cache():
    """drop the resident model of `_cached_model` (the next predict_* call reads plans.json and the checkpoint again)"""
    _MODEL_CACHE.clear()


def _file_stamp(path):
    try:
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)
    except OSError:
        return None


def _model_stamp(model, folds, checkpoint_name):
    """(mtime_ns, size) of plans.json, of every selected fold's <checkpoint_name>.model and of the config files plans.json may name: a
    checkpoint rewritten in place (same plans) must not be served from the cache.  Ranks without checkpoint files stamp what they have."""
    stamp = [_file_stamp(join(model, "plans.json"))]
    try:
        fold_dirs = _fold_dirs(model, folds)
    except OSError:
        fold_dirs = []
    for f in fold_dirs:
        stamp.append((f, _file_stamp(join(model, f, checkpoint_name + ".model"))))
    for extra in sorted(glob.glob(join(model, "*.yaml"))):
        stamp.append((os.path.basename(extra), _file_stamp(extra)))
    return tuple(stamp)


def _cached_model(model, folds, mixed_precision, checkpoint_name):
    """load_model_and_checkpoint_files + load_checkpoint_ram once per (folder, folds, checkpoint, mixed_precision, device, the CF_* knobs in
    force) and per state of the files on disk (`_model_stamp`): predict_from_folder used to rebuild both networks and re-read the checkpoint
    for every patient.  The reference re-reads the checkpoint on every predict_cases call; `clear_model_cache()` forces that here."""
    knobs = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("CF_")))
    key = (os.path.abspath(model), str(folds), checkpoint_name, bool(mixed_precision), torch.cuda.current_device(), knobs)
    stamp = _model_stamp(model, folds, checkpoint_name)
    hit = _MODEL_CACHE.get(key)
    if hit is None or hit[0] != stamp:
        _MODEL_CACHE.clear()                                                     # one model resident at a time
        trainer, params = load_model_and_checkpoint_files(model, folds, mixed_precision=mixed_precision, checkpoint_name=checkpoint_name)
        trainer.load_checkpoint_ram(params[0], False)
        hit = (stamp, trainer, params)
        _MODEL_CACHE[key] = hit
    return hit[1], hit[2]


def _predict_patients(model, cases, folds, save_npz, num_threads_preprocessing, num_threads_nifti_save, do_tta, mixed_precision, all_in_gpu,
                      step_size, checkpoint_name, segmentation_export_kwargs, disable_postprocessing, max_slices=None):
    """predict.py:228-354 + :1008-1110 for a LIST of patients (`cases[i]` = (list_of_lists, output_filenames, ed_index)): the model is loaded
    once, frames are read and preprocessed by a thread pool one group of patients ahead, every group's cropped slices (up to `max_slices`)
    share one device batch, and finished patients are exported by the NIfTI pool while the next group is on the device."""
    import sys
    import time
    from collections import deque
    t_start = time.perf_counter()
    max_slices = max_slices or MAX_SLICES_PER_LAUNCH
    # the calling thread issues ~1500 kernel launches per device batch from Python while up to 32 pool threads read, crop and compress: with
    # the interpreter's default 5 ms switch interval every hand-over of the GIL to a pool thread could stall the launch stream for
    # milliseconds (device time of the 16-patient API bench 6.4 -> 10.4 s once preprocessing overlapped fully, profiles/r03_api_split.md)
    switch_prev = sys.getswitchinterval()
    trainer, params = _cached_model(model, folds, mixed_precision, checkpoint_name)
    timing = {"load_s": time.perf_counter() - t_start, "preprocess_wait_s": 0.0, "preprocess_work_s": 0.0, "device_s": 0.0, "export_wait_s": 0.0,
              "export_work_s": 0.0, "device_batches": 0, "patients": len(cases), "frames": 0, "slices": 0}
    if segmentation_export_kwargs is None:                                       # predict.py:286-296
        exp = trainer.plans.get("segmentation_export_params") or {}
        force_separate_z = exp.get("force_separate_z")
        interpolation_order = exp.get("interpolation_order", 1)
        interpolation_order_z = exp.get("interpolation_order_z", 0)
    else:
        force_separate_z = segmentation_export_kwargs["force_separate_z"]
        interpolation_order = segmentation_export_kwargs["interpolation_order"]
        interpolation_order_z = segmentation_export_kwargs["interpolation_order_z"]
    orders = []
    for list_of_lists, output_filenames, ed_index in cases:
        assert len(list_of_lists) == len(output_filenames)
        for o in output_filenames:
            for sub in ("Segmentation", "Flow", "Registered"):
                os.makedirs(join(os.path.dirname(o), sub), exist_ok=True)
        T = len(list_of_lists)
        orders.append(list(range(ed_index, T)) + list(range(0, ed_index)))      # ED first (put_ed_first, predict.py:1165-1193)

    import itertools
    import threading
    tls = threading.local()
    stream_ids = itertools.count()

    def pre_one(files):
        # every preprocessing thread issues its (small) device kernels on a HIP stream of its own: on the default stream they -- and the
        # host read-backs between them -- queued behind the seconds-long network batch of the main thread.  The streams are taken from a
        # process-wide pool (thread k of every call gets stream k): torch's caching allocator keeps one pool of blocks per stream, so fresh
        # streams in every call meant fresh hipMalloc calls -- each a device-wide synchronisation -- under the running network batch
        t0 = time.perf_counter()
        if not hasattr(tls, "stream"):
            torch.cuda.set_device(trainer.device)                                # the current device is per thread (a new thread starts on GPU 0)
            tls.stream = _pooled_stream(trainer.device, next(stream_ids))
        with torch.cuda.stream(tls.stream):
            r = trainer.preprocess_patient(files)                                # predict.py:302
            tls.stream.synchronize()
        return r, time.perf_counter() - t0

    pre_pool = ThreadPool(max(1, num_threads_preprocessing))
    pool = ThreadPool(max(1, num_threads_nifti_save))
    submitted = deque()                                                          # (case index, [async results per frame])
    nxt = 0

    def submit_more(lookahead):
        nonlocal nxt
        while nxt < len(cases) and len(submitted) < lookahead:
            lol = cases[nxt][0]
            submitted.append((nxt, [pre_pool.apply_async(pre_one, (lol[i],)) for i in orders[nxt]]))
            nxt += 1

    finishing = deque()                                                          # exports in flight: (seg_paths, reg_paths, jobs, output files)
    sys.setswitchinterval(GIL_SWITCH_INTERVAL)                                   # restored in the finally below
    try:
        # slices of a patient are only known after preprocessing; groups are filled greedily in patient order.  The pool runs `ahead` patients
        # in front of the collector: at least one whole device batch more than the group being assembled, so that the frames of the NEXT
        # group are read and cropped while this one is on the device (with 4 the second half of the next group was only submitted after
        # the device batch had finished: 3.4 of 10.5 s of the 16-patient API bench were spent waiting for it, profiles/r03_api_split.md)
        ahead = 8
        if PREFETCH_ALL:
            # read and preprocess the WHOLE request before the first device batch: the network kernels run ~35 % longer while the small
            # preprocessing kernels and their read-backs share the GPU (profiles/r03_api_split.md), and reading is 1-2 s per 16 patients
            ahead = len(cases)
            submit_more(ahead)
            t0 = time.perf_counter()
            sampler = _StackSampler() if os.environ.get("CF_API_PROFILE", "0") == "2" else None
            for _ci, asyncs in submitted:
                for a in asyncs:
                    a.wait()
            if sampler:
                sampler.report("preprocessing threads while the request is read (%.2f s)" % (time.perf_counter() - t0))
            timing["preprocess_wait_s"] += time.perf_counter() - t0
        submit_more(ahead)
        carry = None
        # the first device batch is small (FIRST_BATCH_SLICES) and the cap doubles from batch to batch: the device starts as soon as two
        # or so patients are read instead of waiting for a full batch of 64 slices, and the later patients are preprocessed behind it
        cap = max_slices if PREFETCH_ALL else min(max_slices, FIRST_BATCH_SLICES)
        while submitted or carry is not None:
            group, nslices = [], 0
            while carry is not None or submitted:
                if carry is None:
                    ci, asyncs = submitted.popleft()
                    t0 = time.perf_counter()
                    got = [a.get() for a in asyncs]
                    timing["preprocess_wait_s"] += time.perf_counter() - t0
                    timing["preprocess_work_s"] += sum(g[1] for g in got)
                    carry = (ci, [g[0] for g in got])
                    submit_more(ahead)
                z = carry[1][0][0].shape[1]
                if group and nslices + z > cap:
                    break
                group.append(carry)
                nslices += z
                carry = None
            t0 = time.perf_counter()
            unl = [np.stack([p_[0] for p_ in pre]) + 1e-8 for _ci, pre in group]      # predict.py:1025
            print("predicting %d patient(s), %d slices in one device batch" % (len(group), nslices))
            cap = min(max_slices, 2 * cap)
            ahead = max(ahead, 2 * len(group) + 2)
            submit_more(ahead)
            # the crop-space copies only when the voxelmorph_saver tree is being written; the `raw` tensor (frames + crop-space flow) the
            # reference returns for its trainer's plots is not consumed by the exporter
            # the probabilities come to the host only if the exporter needs them: for the npz, or to resample them back to the size
            # before the preprocessing's resampling (segmentation_export.py:84-127); otherwise it writes the device arg-max
            tf_ = list(trainer.plans["transpose_forward"])
            resampled = any(tuple(p_[0].shape[1:]) != tuple(np.array(p_[2]["size_after_cropping"])[tf_]) for _ci, pre in group for p_ in pre)
            results = trainer.predict_patients_flow(unl, do_mirroring=do_tta, mirror_axes=trainer.data_aug_params["mirror_axes"],
                                                    return_crop=_VOXELMORPH_RAW is not None, want_raw=False, want_softmax=bool(save_npz or resampled))
            torch.cuda.synchronize()
            timing["device_s"] += time.perf_counter() - t0
            timing["device_batches"] += 1
            timing["slices"] += nslices
            for (ci, pre), res in zip(group, results):
                outs = [cases[ci][1][i] for i in orders[ci]]
                timing["frames"] += len(outs)
                sp, _fp, rp, jobs = _export_flow_patient(res, trainer, outs, [p_[2] for p_ in pre], interpolation_order, force_separate_z,
                                                         interpolation_order_z, save_npz, pool)
                finishing.append((sp, rp, jobs, outs))
            while len(finishing) > 2 * max(1, len(group)):                       # bound the host memory held by queued exports
                t0 = time.perf_counter()
                sp, rp, jobs, outs = finishing.popleft()
                timing["export_work_s"] += _finish_flow_patient(sp, rp, jobs, outs, disable_postprocessing, model)
                timing["export_wait_s"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        while finishing:
            sp, rp, jobs, outs = finishing.popleft()
            timing["export_work_s"] += _finish_flow_patient(sp, rp, jobs, outs, disable_postprocessing, model)
        timing["export_wait_s"] += time.perf_counter() - t0
    except BaseException:
        # a failed batch must not wait for every queued preprocessing / export job: drop them
        pre_pool.terminate()
        pool.terminate()
        raise
    finally:
        pre_pool.close()                                                         # (no-ops after terminate())
        pool.close()
        pre_pool.join()
        pool.join()
        sys.setswitchinterval(switch_prev)
    timing["total_s"] = time.perf_counter() - t_start
    if API_PROFILE:
        timing.update({"device_" + k: v for k, v in DEVICE_SPLIT.items()})
        DEVICE_SPLIT.clear()
    LAST_TIMING.clear()
    LAST_TIMING.update(timing)
    return [[(_subfolder_path(o, "Segmentation"), _subfolder_path(o, "Flow")[:-7] + ".npz", _subfolder_path(o, "Registered")) for o in c[1]]
            for c in cases]


def predict_cases(model, list_of_lists, output_filenames, folds, save_npz, num_threads_preprocessing, num_threads_nifti_save,
                  segs_from_prev_stage=None, do_tta=True, mixed_precision=True, overwrite_existing=False, all_in_gpu=False,
                  step_size=0.5, checkpoint_name="model_final_checkpoint", segmentation_export_kwargs=None,
                  disable_postprocessing=False, ed_index=0):
    """predict.py:228-354 for ONE patient: `list_of_lists[t]` = the modality files of frame t.  All frames form the cine
    sequence; frame `ed_index` is rotated to the front for the ED-anchored recurrence (put_ed_first, :1165-1193)."""
    assert len(list_of_lists) == len(output_filenames)
    if segs_from_prev_stage is not None:
        assert len(segs_from_prev_stage) == len(output_filenames)
    return _predict_patients(model, [(list_of_lists, output_filenames, ed_index)], folds, save_npz, num_threads_preprocessing, num_threads_nifti_save,
                             do_tta, mixed_precision, all_in_gpu, step_size, checkpoint_name, segmentation_export_kwargs, disable_postprocessing)[0]


def predict_cases_fast(model, list_of_lists, output_filenames, folds, num_threads_preprocessing, num_threads_nifti_save,
                       segs_from_prev_stage=None, do_tta=True, mixed_precision=True, overwrite_existing=False, all_in_gpu=False,
                       step_size=0.5, checkpoint_name="model_final_checkpoint", segmentation_export_kwargs=None,
                       disable_postprocessing=False):
    """predict.py:356-501 ("fast": no resampled-softmax npz).  Everything already stays on the GPU here, so this is
    predict_cases with save_npz=False."""
    return predict_cases(model, list_of_lists, output_filenames, folds, False, num_threads_preprocessing, num_threads_nifti_save,
                         segs_from_prev_stage, do_tta, mixed_precision, overwrite_existing, all_in_gpu, step_size, checkpoint_name,
                         segmentation_export_kwargs, disable_postprocessing)


def predict_cases_fastest(model, list_of_lists, output_filenames, folds, num_threads_preprocessing, num_threads_nifti_save,
                          segs_from_prev_stage=None, do_tta=True, mixed_precision=True, overwrite_existing=False, all_in_gpu=False,
                          step_size=0.5, checkpoint_name="model_final_checkpoint", disable_postprocessing=False):
    """predict.py:504-626 ("fastest": argmax on the device, nearest-neighbour export)."""
    return predict_cases(model, list_of_lists, output_filenames, folds, False, num_threads_preprocessing, num_threads_nifti_save,
                         segs_from_prev_stage, do_tta, mixed_precision, overwrite_existing, all_in_gpu, step_size, checkpoint_name,
                         {"force_separate_z": None, "interpolation_order": 0, "interpolation_order_z": 0}, disable_postprocessing)


def predict_from_folder(model, input_folder, output_folder, folds, save_npz, num_threads_preprocessing, num_threads_nifti_save,
                        lowres_segmentations, part_id, num_parts, tta, mixed_precision=True, overwrite_existing=True, mode="normal",
                        overwrite_all_in_gpu=None, step_size=0.5, checkpoint_name="model_final_checkpoint",
                        segmentation_export_kwargs=None, disable_postprocessing=False):
    """predict.py:665-780.  Patients are sharded `patients[part_id::num_parts]` (one process per GPU); every patient of
    the shard is processed.  (set_voxelmorph_raw / the CLI's --voxelmorph_raw additionally produce the voxelmorph_saver input tree;
    the argument list itself is the reference's, name for name.)"""
    os.makedirs(output_folder, exist_ok=True)
    assert os.path.isfile(join(model, "plans.json")), "Folder with saved model weights must contain a plans.json file"
    shutil.copy(join(model, "plans.json"), output_folder)
    with open(join(model, "plans.json")) as f:
        expected_num_modalities = json.load(f)["num_modalities"]
    if mode not in ("normal", "fast", "fastest"):
        raise ValueError("unrecognized mode. Must be normal, fast or fastest")
    patients = sorted(p for p in os.listdir(input_folder) if os.path.isdir(join(input_folder, p)))
    # one process per GPU under torchrun (RANK / WORLD_SIZE): the reference's partition with part_id = rank, num_parts = world unless the
    # caller partitions explicitly (predict.py:743, :806-821)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and num_parts == 1:
        part_id, num_parts = int(os.environ.get("RANK", "0")), world
    shard, cases = patients[part_id::num_parts], []
    for patient in shard:
        current_input_folder = join(input_folder, patient)
        current_output_folder = join(output_folder, patient)
        for sub in ("Flow", "Registered", "Segmentation"):
            os.makedirs(join(current_output_folder, sub), exist_ok=True)
        case_ids = check_input_folder_and_return_caseIDs(current_input_folder, expected_num_modalities)
        output_files = [join(current_output_folder, i + ".nii.gz") for i in case_ids]
        all_files = subfiles(current_input_folder, suffix=".nii.gz", join_=False, sort=True)
        list_of_lists = [[join(current_input_folder, i) for i in all_files if i[:len(j)].startswith(j) and len(i) == (len(j) + 12)]
                         for j in case_ids]
        ed_index = 0
        csv_path = join(current_input_folder, patient + ".csv")                  # predict.py:700, :1196-1198
        if os.path.isfile(csv_path):
            ed_index = get_ed_es_indices(csv_path)[0]
        cases.append((list_of_lists, output_files, ed_index))
    seg_exp = segmentation_export_kwargs
    if mode != "normal":
        assert save_npz is False                                                 # predict.py:755, :771
    if mode == "fastest":                                                        # predict.py:504-626: nearest-neighbour export
        seg_exp = {"force_separate_z": None, "interpolation_order": 0, "interpolation_order_z": 0}
    if not cases and world > 1:
        _cached_model(model, folds, mixed_precision, checkpoint_name)            # an empty shard still takes part in the weight broadcast
    res = _predict_patients(model, cases, folds, save_npz, num_threads_preprocessing, num_threads_nifti_save, tta, mixed_precision,
                            bool(overwrite_all_in_gpu), step_size, checkpoint_name, seg_exp, disable_postprocessing) if cases else []
    return dict(zip(shard, res))


def main(argv=None):
    """CLI flags of predict.py:782-858 / predict_simple.py:34-131."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-i", "--input_folder", required=True)
    parser.add_argument("-o", "--output_folder", required=True)
    parser.add_argument("-m", "--model_output_folder", required=True)
    parser.add_argument("-f", "--folds", nargs="+", default="None")
    parser.add_argument("-z", "--save_npz", required=False, action="store_true")
    parser.add_argument("-l", "--lowres_segmentations", required=False, default="None")
    parser.add_argument("--part_id", type=int, required=False, default=0)
    parser.add_argument("--num_parts", type=int, required=False, default=1)
    parser.add_argument("--num_threads_preprocessing", required=False, default=6, type=int)
    parser.add_argument("--num_threads_nifti_save", required=False, default=2, type=int)
    parser.add_argument("--tta", required=False, type=int, default=1)
    parser.add_argument("--disable_tta", required=False, default=False, action="store_true")
    parser.add_argument("--overwrite_existing", required=False, type=int, default=1)
    parser.add_argument("--mode", type=str, default="normal", required=False)
    parser.add_argument("--all_in_gpu", type=str, default="None", required=False)
    parser.add_argument("--step_size", type=float, default=0.5, required=False)
    parser.add_argument("--disable_mixed_precision", default=False, action="store_true", required=False)
    parser.add_argument("-chk", default="model_final_checkpoint", required=False)
    parser.add_argument("--voxelmorph_raw", default=None, required=False, help="also write <dir>/Raw/{Registered,Segmentation,Flow}/<patient>/ "
                        "(crop-space predictions, the input of voxelmorph_saver_*)")
    parser.add_argument("--voxelmorph_pkl", default=None, required=False, help="folder for the per-file .pkl properties (default <voxelmorph_raw>/pkl)")
    a = parser.parse_args(argv)
    folds = a.folds if a.folds != "None" and a.folds != ["None"] else None
    if isinstance(folds, list):
        folds = [int(i) if i != "all" else i for i in folds]
    all_in_gpu = None if a.all_in_gpu == "None" else a.all_in_gpu == "True"
    tta = bool(a.tta) and not a.disable_tta
    if a.voxelmorph_raw is not None:
        set_voxelmorph_raw(a.voxelmorph_raw, a.voxelmorph_pkl)
    return predict_from_folder(a.model_output_folder, a.input_folder, a.output_folder, folds, a.save_npz, a.num_threads_preprocessing,
                               a.num_threads_nifti_save, None, a.part_id, a.num_parts, tta, mixed_precision=not a.disable_mixed_precision,
                               overwrite_existing=bool(a.overwrite_existing), mode=a.mode, overwrite_all_in_gpu=all_in_gpu,
                               step_size=a.step_size, checkpoint_name=a.chk)


if __name__ == "__main__":
    main()
