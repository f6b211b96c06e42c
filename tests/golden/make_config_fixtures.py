#!/usr/bin/env python3
"""Generate tests/golden/configs.json: the VALUES of the four reference YAML configs on the hot path plus what the reference's own classes
make of them, measured in the build container.

    cd /tmp && python /root/repo/tests/golden/make_config_fixtures.py

For nnunet/{raft_config,video,successive,adversarial_acdc}.yaml the script
  * parses the file (pyyaml safe loader) and stores the mapping as JSON -- data, not the file: comments and layout are gone;
  * calls the reference's constructors with the keyword map of the reference's builders (nnunet/lib/training_utils.py:1460-1537,
    :1256-1286, :1938-1996; that module itself cannot be imported, 20 of its imports are absent from the snapshot), every value read
    as config[key] exactly like there, and records the parameter count, the number of state-dict entries and a digest of the
    (name, shape) list.  `prediction: False` is supplied for raft_config.yaml, which lacks the key the builder reads;
  * records the first key a builder would fail on (`missing_key`).
tests/test_config.py feeds the stored values through cineflow.config and checks the resulting state_shapes() against these numbers.
"""
import hashlib
import json
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cardiac-segmentation-optical-flow_amd"))

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import models as OM  # noqa: E402
from cineflow import config as C  # noqa: E402

REF = os.path.join(_ref_import.REFERENCE_ROOT, "nnunet")


def digest(sd, skip=("grid",)):
    """count / entries / sha1 over sorted (name, shape) of a state dict (grids of SpatialTransformer excluded: input-size buffers)"""
    items = sorted((k, tuple(v.shape)) for k, v in sd.items() if not k.endswith(skip))
    n = sum(int(torch.tensor(s).prod()) if len(s) else 1 for _, s in items)
    h = hashlib.sha1(json.dumps(items).encode()).hexdigest()
    return {"param_count": n, "entries": len(items), "names_shapes_sha1": h}


def first_missing(config, keys):
    for _, k in keys:
        if k not in config:
            return k
    return None


def main():
    import nnunet.lib.raft as ref_raft_stub
    ref_raft_stub.CorrVolume = OM.CorrVolume              # source absent; parameter-free, so the counts are the reference's own
    import nnunet.network_architecture.SegFlowGaussian as ref_sfg
    import nnunet.network_architecture.Optical_flow_model_successive as ref_succ
    ref_succ.NCC = lambda reduction=None: None
    from nnunet.network_architecture.MTL_model import MTLmodel
    from nnunet.lib.utils import ConvBlocks2DGroup
    torch.cuda.FloatTensor = torch.FloatTensor            # SegFlowGaussian.py:350 passes it as a dtype tag

    out = {}
    for name in ("raft_config", "video", "successive", "adversarial_acdc"):
        with open(os.path.join(REF, name + ".yaml")) as f:
            cfg = yaml.safe_load(f)
        rec = {"values": cfg}
        if name in ("raft_config", "video"):
            rec["missing_key"] = first_missing(cfg, C._SEGFLOW_KEYS)
            full = C.with_defaults(cfg, prediction=False)
            kw = {n: full[k] for n, k in C._SEGFLOW_KEYS}
            ref = ref_sfg.SegFlowGaussian(image_size=256, log_function=print, **kw)
            rec["image_size"] = 256
            rec["reference"] = digest(ref.state_dict())
        elif name == "successive":
            rec["missing_key"] = first_missing(cfg, C._SUCCESSIVE_KEYS)

            def build(nb_channels):
                kw = {n: (list(cfg[k]) if isinstance(cfg[k], list) else cfg[k]) for n, k in C._SUCCESSIVE_KEYS}
                return ref_succ.OpticalFlowModelSuccessive(image_size=256, log_function=print, nb_channels=nb_channels, backward=False,
                                                           segmentation=False, dot_multiplier=2, **kw)
            ref = ref_succ.ModelWrap(build(1), build(6), do_ds=False, motion_from_ed=cfg["motion_from_ed"], backward=False, segmentation=False,
                                     no_error=cfg["no_error"], use_label=False)
            rec["image_size"] = 256
            rec["reference"] = digest(ref.state_dict())
        else:
            rec["missing_key"] = first_missing(cfg, C._MTL_KEYS)
            kw = {n: cfg[k] for n, k in C._MTL_KEYS}
            ref = MTLmodel(conv_layer=ConvBlocks2DGroup, num_classes=2, log_function=None, norm=getattr(torch.nn, cfg["norm"]), middle=False,
                           processor=None, window_size=7, image_size=224, add_absolute_pos=False, init_weights=None, **kw)
            rec["image_size"], rec["window_size"], rec["num_classes"] = 224, 7, 2
            rec["reference"] = digest(ref.state_dict(), skip=("grid", "num_batches_tracked", "relative_position_index", "attn_mask"))
        print("%-18s missing key: %-12s reference: %s" % (name, rec["missing_key"], rec["reference"]))
        out[name] = rec
    with open(os.path.join(HERE, "configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote configs.json (%.1f KiB)" % (os.path.getsize(os.path.join(HERE, "configs.json")) / 1024))


if __name__ == "__main__":
    main()
