#!/usr/bin/env python3
"""Generate tests/golden/mtl_*.npz by RUNNING THE REFERENCE's MTLmodel in the build container and pin oracle/mtl.py.

    cd /tmp && python /root/repo/tests/golden/make_golden_mtl.py

`nnunet.network_architecture.MTL_model.MTLmodel` is imported in place (inert import stubs for the absent third-party packages,
tests/golden/_ref_import.py) and built with the values of nnunet/adversarial_acdc.yaml at reduced width (image 64, window 8,
in_dims [1,16,32], out_encoder_dims [8,16,32], heads [2,2,4]); `add_absolute_pos=False`, `init_weights=None` are the two required
arguments the reference's own build function never passes.  Parameters AND BatchNorm running statistics get the seeded fill, the
reference's state dict is loaded into the oracle with strict=True, both run on seeded inputs in eval mode.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cardiac-segmentation-optical-flow_amd"))

import _ref_import  # noqa: E402

_ref_import.install()

from oracle import mtl as OMTL  # noqa: E402
from cineflow.weights import fill_module_  # noqa: E402

REPORT = []


def pin(name, ref, ora, tol=0.0):
    ref, ora = np.asarray(ref, dtype=np.float64), np.asarray(ora, dtype=np.float64)
    assert ref.shape == ora.shape, (name, ref.shape, ora.shape)
    d = float(np.abs(ref - ora).max())
    REPORT.append((name, d, tol))
    print("  oracle vs reference %-46s max|diff| = %.3e (tol %.1e)" % (name, d, tol))
    assert d <= tol, name


def build_reference(image_size, window_size, num_classes, in_dims, out_dims, heads, bottleneck_heads):
    from nnunet.network_architecture.MTL_model import MTLmodel
    from nnunet.lib.utils import ConvBlocks2DGroup
    return MTLmodel(device="cpu", conv_layer=ConvBlocks2DGroup, num_classes=num_classes, transformer_bottleneck=True, separability=False,
                    adversarial_loss=False, log_function=None, asymmetric_unet=True, norm=torch.nn.BatchNorm2d, affinity=False,
                    add_extra_bottleneck_blocks=True, middle=False, filter_skip_co_segmentation=True, directional_field=False, classification=False,
                    batch_size=4, uncertainty_weighting=False, reconstruction=False, reconstruction_skip=False, proj="linear", processor=None,
                    shortcut=False, use_conv_mlp=True, similarity_down_scale=8, concat_spatial_cross_attention=False, encoder_attention_type=None,
                    spatial_cross_attention_num_heads=heads, merge="linear", out_encoder_dims=out_dims, swin_abs_pos=False,
                    patch_size=[image_size, image_size], window_size=window_size, in_dims=in_dims, deep_supervision=True, bottleneck="swin",
                    drop_path_rate=0.0, image_size=image_size, conv_depth=[2, 2, 2], transformer_depth=[], num_heads=[],
                    bottleneck_heads=bottleneck_heads, num_bottleneck_layers=1, rpe_mode="bias", rpe_contextual_tensor="qkv",
                    add_absolute_pos=False, init_weights=None)


def main():
    from nnunet.lib.swin_cross_attention import SwinFilterBlock
    g = torch.Generator().manual_seed(91)
    # ---- the skip filter alone: 32x32 map, window 8 -> 16 windows, the second block shifted by 4 (masked windows at the border)
    ref = fill_module_(SwinFilterBlock(in_dim=16, out_dim=16, input_resolution=(32, 32), num_heads=2, norm=torch.nn.BatchNorm2d, device="cpu",
                                       rpe_mode="bias", rpe_contextual_tensor="qkv", window_size=8, depth=2, add_absolute_pos=False, init_weights=None), 61).eval()
    ora = OMTL.SwinFilterBlock(16, 16, (32, 32), 2, 8).eval()
    ora.load_state_dict(ref.state_dict(), strict=True)
    x, skip = torch.randn(2, 16, 32, 32, generator=g), torch.randn(2, 16, 32, 32, generator=g)
    with torch.no_grad():
        r, o = ref(x, skip), ora(x, skip)
    pin("SwinFilterBlock (window 8, shifted second block)", r, o, 1e-6)
    np.savez_compressed(os.path.join(HERE, "mtl_filter.npz"), x=x.numpy(), skip=skip.numpy(), out=r.numpy())
    # ---- window 7 on a 28x28 map (the ACDC setting: image 224, window 7)
    ref = fill_module_(SwinFilterBlock(in_dim=8, out_dim=8, input_resolution=(28, 28), num_heads=2, norm=torch.nn.BatchNorm2d, device="cpu",
                                       rpe_mode="bias", rpe_contextual_tensor="qkv", window_size=7, depth=2, add_absolute_pos=False, init_weights=None), 62).eval()
    ora = OMTL.SwinFilterBlock(8, 8, (28, 28), 2, 7).eval()
    ora.load_state_dict(ref.state_dict(), strict=True)
    x, skip = torch.randn(1, 8, 28, 28, generator=g), torch.randn(1, 8, 28, 28, generator=g)
    with torch.no_grad():
        r, o = ref(x, skip), ora(x, skip)
    pin("SwinFilterBlock (window 7)", r, o, 1e-6)
    np.savez_compressed(os.path.join(HERE, "mtl_filter7.npz"), x=x.numpy(), skip=skip.numpy(), out=r.numpy())
    # ---- the whole network, 2 classes (the cropping network) and 4 classes (the segmenter)
    for tag, ncls, seed in (("crop", 2, 63), ("seg", 4, 64)):
        ref = fill_module_(build_reference(64, 8, ncls, [1, 16, 32], [8, 16, 32], [2, 2, 4], 8), seed).eval()
        ref.do_ds = False
        ora = OMTL.MTLmodel(64, 8, ncls, [1, 16, 32], [8, 16, 32], [2, 2, 2], [2, 2, 4], 8, 1).eval()
        missing = ora.load_state_dict(ref.state_dict(), strict=True)
        x = torch.randn(2, 1, 64, 64, generator=g)
        with torch.no_grad():
            r, o = ref(x)["pred"], ora(x)["pred"]
        pin("MTLmodel %d classes (adversarial_acdc.yaml, reduced)" % ncls, r, o, 2e-6)
        np.savez_compressed(os.path.join(HERE, "mtl_%s.npz" % tag), x=x.numpy(), pred=r.numpy())
    with open(os.path.join(HERE, "PIN_REPORT_mtl.txt"), "w") as f:
        f.write("oracle/mtl.py vs the reference (tests/golden/make_golden_mtl.py)\n")
        for name, d, tol in REPORT:
            f.write("%-58s max|diff| = %.3e  (tol %.1e)\n" % (name, d, tol))
    print("%d pins" % len(REPORT))


if __name__ == "__main__":
    main()
