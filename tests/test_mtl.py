"""MTLmodel: oracle vs the reference's own outputs (tests/golden/mtl_*.npz, make_golden_mtl.py) on the CPU,
device path vs the same vectors on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RED = dict(in_dims=[1, 16, 32], out_encoder_dims=[8, 16, 32], conv_depth=[2, 2, 2], spatial_cross_attention_num_heads=[2, 2, 4], bottleneck_heads=8)


def _gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_oracle_mtl_matches_reference_outputs():
    from oracle import mtl as OMTL
    from cineflow.weights import fill_module_
    for tag, ncls, seed in (("crop", 2, 63), ("seg", 4, 64)):
        g = _gold("mtl_" + tag)
        m = fill_module_(OMTL.MTLmodel(64, 8, ncls, **RED), seed).eval()
        with torch.no_grad():
            out = m(torch.from_numpy(g["x"]))["pred"]
        assert float((out - torch.from_numpy(g["pred"])).abs().max()) <= 1e-6
    g = _gold("mtl_filter")
    f = fill_module_(OMTL.SwinFilterBlock(16, 16, (32, 32), 2, 8), 61).eval()
    with torch.no_grad():
        assert float((f(torch.from_numpy(g["x"]), torch.from_numpy(g["skip"])) - torch.from_numpy(g["out"])).abs().max()) <= 1e-6


def test_mtl_state_dict_layout_matches_oracle():
    """same parameter / running-statistics names and shapes as the oracle, which loads the reference's state dict with strict=True"""
    from cineflow.mtl import MTLmodel
    from oracle import mtl as OMTL
    mine = MTLmodel(64, 8, 2, **RED).state_shapes()
    ref = {k: tuple(v.shape) for k, v in OMTL.MTLmodel(64, 8, 2, **RED).state_dict().items()
           if not k.endswith(("num_batches_tracked", "relative_position_index", "attn_mask"))}
    assert mine == ref
    # full width (adversarial_acdc.yaml: in [1,128,256], out [64,128,256], heads [2,4,8], d_model 512)
    full = MTLmodel(224, 7, 4).state_shapes()
    assert full["bottleneck.layers.0.self_attn.in_proj_weight"] == (1536, 512) and full["decoder.layers.2.blocks.0.0.weight"] == (4, 128, 3, 3)
    assert full["decoder.encoder_skip_layers.0.blocks.1.cross_attn.relative_position_bias_table"] == (169, 8)


@pytest.mark.gpu
def test_swin_filter_block_vs_reference(dev):
    from cineflow.mtl import SwinFilterBlock
    from cineflow.weights import seeded_state_dict
    for name, args, seed in (("mtl_filter", (16, 16, (32, 32), 2, 8), 61), ("mtl_filter7", (8, 8, (28, 28), 2, 7), 62)):
        g = _gold(name)
        m = SwinFilterBlock(*args)
        m.load_state_dict(seeded_state_dict(m.state_shapes(), seed), dev)
        out = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["skip"]).to(dev)).cpu()
        assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5, name


@pytest.mark.gpu
def test_mtl_model_vs_reference(dev):
    from cineflow.mtl import MTLmodel
    from cineflow.weights import seeded_state_dict
    for tag, ncls, seed in (("crop", 2, 63), ("seg", 4, 64)):
        g = _gold("mtl_" + tag)
        m = MTLmodel(64, 8, ncls, **RED)
        m.load_state_dict(seeded_state_dict(m.state_shapes(), seed), dev)
        out = m(torch.from_numpy(g["x"]).to(dev))["pred"].cpu()
        ref = torch.from_numpy(g["pred"])
        assert float((out - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())), tag


@pytest.mark.gpu
def test_attention_ragged_token_count(dev):
    """28 x 28 = 784 tokens (the bottleneck of a 224 x 224 image): padded to 800 on the host, the 16 padded keys masked in the kernel"""
    from cineflow import ops
    g = torch.Generator().manual_seed(5)
    for heads, hd in ((8, 64), (8, 8)):
        C, N = heads * hd, 784
        q, k, v = (torch.randn(2, C, N, generator=g) for _ in range(3))
        out = ops.attention_cf(q.to(dev), k.to(dev), v.to(dev), heads).cpu()
        qh, kh, vh = (t.view(2, heads, hd, N).transpose(2, 3).double() for t in (q, k, v))
        ref = (torch.softmax(qh @ kh.transpose(2, 3) / np.sqrt(hd), dim=-1) @ vh).transpose(2, 3).reshape(2, C, N)
        assert float((out.double() - ref).abs().max()) <= 2e-5


@pytest.mark.gpu
def test_mtl_full_width_and_tta_with_processor(dev):
    """adversarial_acdc.yaml at full width on a 224 x 224 image (window 7, 784-token bottleneck) against the oracle, then the
    inference wrapper of MTL_model.py:816-936 with a Processor: crop around the cropping network's centroid, flip TTA, un-crop."""
    from cineflow.mtl import MTLmodel
    from cineflow.inference import Processor
    from cineflow.weights import seeded_state_dict, fill_module_
    from oracle import mtl as OMTL
    from oracle import models as OM
    m = MTLmodel(224, 7, 4)
    m.load_state_dict(seeded_state_dict(m.state_shapes(), 71), dev)
    o = fill_module_(OMTL.MTLmodel(224, 7, 4), 71).eval()
    x = torch.randn(1, 1, 224, 224, generator=torch.Generator().manual_seed(72))
    with torch.no_grad():
        ref = o(x)["pred"]
    out = m(x.to(dev))["pred"].cpu()
    assert float((out - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))
    # wrapper at reduced width: cropping network = a 2-class MTLmodel, segmenter = a 4-class one on the 64 x 64 crop of a 96 x 96 image
    crop_net = MTLmodel(96, 8, 2, **RED)
    crop_net.load_state_dict(seeded_state_dict(crop_net.state_shapes(), 73), dev)
    seg = MTLmodel(64, 8, 4, processor=Processor(64, 96, crop_net), **RED)
    seg.load_state_dict(seeded_state_dict(seg.state_shapes(), 74), dev)
    ocrop = fill_module_(OMTL.MTLmodel(96, 8, 2, **RED), 73).eval()
    oseg = fill_module_(OMTL.MTLmodel(64, 8, 4, **RED), 74).eval()
    oproc = OM.Processor(64, 96, ocrop)
    xb = torch.randn(2, 1, 96, 96, generator=torch.Generator().manual_seed(75)) * 20 + 60
    got = seg.mirror_and_predict_2d(xb.to(dev), normalize=True).cpu()
    from oracle import ops as OO
    with torch.no_grad():
        want = []
        for b in range(2):
            cen, _ = oproc.preprocess_no_registration(xb[b][None].clone())
            c, pn = oproc.crop_and_pad(xb[b][None], cen)
            c = OO.normalize_intensity(c[0])[None]
            p = OM.mirror_and_predict_2d(type("N", (), {"num_classes": 4, "__call__": lambda s, t: oseg(t)["pred"]})(), c)
            want.append(oproc.uncrop_no_registration(p, pn[None])[0])
        want = torch.stack(want)
    assert got.shape == want.shape == (2, 4, 96, 96)
    assert float((got - want).abs().max()) <= 5e-5
