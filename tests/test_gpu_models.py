"""GPU parity tests of the module/model level against the golden vectors (reference outputs) and the oracle.

Bars (BASELINE.json north_star): mean flow EPE <= 1e-4 px, Dice of propagated labels within 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RED = dict(in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32])
S = 64


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def check(a, b, tol, what=""):
    d = maxdiff(a, b)
    assert d <= tol, "%s max|diff| %.3e > %.1e" % (what, d, tol)


def load(mod, seed, dev):
    from cineflow.weights import seeded_state_dict
    mod.load_state_dict(seeded_state_dict(mod.state_shapes(), seed), dev)
    return mod


def randn(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_convgru(dev, golden):
    from cineflow.nn import ConvGRUCell
    g = golden("convgru")
    m = load(ConvGRUCell((8, 8), 32, 32), 1, dev)
    check(m(T(g["x"]).to(dev), T(g["h"]).to(dev)), g["out"], 1e-5)


@pytest.mark.parametrize("tag,kw", [("res_s1", dict(in_dim=6, out_dim=16, nb_blocks=1, residual=True)),
                                    ("res_s2", dict(in_dim=16, out_dim=32, nb_blocks=1, residual=True, stride=2)),
                                    ("same", dict(in_dim=16, out_dim=16, nb_blocks=1, residual=True)),
                                    ("nores", dict(in_dim=16, out_dim=8, nb_blocks=1, residual=False)),
                                    ("single", dict(in_dim=8, out_dim=16, nb_blocks=1, residual=True, nb_conv=1))])
def test_convblocks(dev, golden, tag, kw):
    from cineflow.nn import ConvBlocks2DGroupLegacy
    g = golden("convblock_" + tag)
    m = load(ConvBlocks2DGroupLegacy(**kw), 2, dev)
    check(m(T(g["x"]).to(dev)), g["out"], 2e-5)


def test_patch_expand_merge(dev, golden):
    from cineflow.nn import PatchExpand2DGroup, PatchMerging2DGroup
    g = golden("patchexpand")
    check(load(PatchExpand2DGroup(32, 16), 3, dev)(T(g["x"]).to(dev)), g["out"], 2e-5)
    g = golden("patchmerging")
    check(load(PatchMerging2DGroup(8, 16), 3, dev)(T(g["x"]).to(dev)), g["out"], 2e-5)


def test_encoders_decoder(dev, golden):
    from cineflow.nn import Encoder2D, Decoder2D
    g = golden("encoder2d")
    m = load(Encoder2D(d_model=32, conv_depth=[1, 1, 1], in_dims=RED["in_dims"], out_dims=RED["out_encoder_dims"], nb_conv=2,
                       extra_block=True, residual=True, downsample_conv=2), 4, dev)
    f, sk = m(T(g["x"]).to(dev))
    check(f, g["feat"], 5e-5)
    for i in range(3):
        check(sk[i], g["skip%d" % i], 5e-5)
    g = golden("encoder_ma")
    m = load(Encoder2D(d_model=32, conv_depth=[1, 1, 1], in_dims=[2, 16, 32], out_dims=RED["out_encoder_dims"], nb_conv=2,
                       extra_block=False, residual=True, downsample_conv=2, motion_appearance=True), 5, dev)
    a, mo, sk = m(T(g["x"]).to(dev))
    check(a, g["app"], 5e-5)
    check(mo, g["motion"], 5e-5)
    g = golden("encoder2d_succ")
    m = load(Encoder2D(d_model=64, conv_depth=[1, 1, 1], in_dims=RED["in_dims"], out_dims=RED["out_encoder_dims"], nb_conv=2,
                       extra_block=False, residual=False, downsample_conv=1), 6, dev)
    f, sk = m(T(g["x"]).to(dev))
    check(f, g["feat"], 5e-5)
    check(sk[2], g["skip2"], 5e-5)
    g = golden("decoder2d")
    m = load(Decoder2D(d_model=32, dot_multiplier=2, conv_depth=[1, 1, 1], in_encoder_dims=[32, 16, 4], out_encoder_dims=[32, 16, 8],
                       num_classes=2, nb_conv=2, residual=True), 7, dev)
    check(m(T(g["x"]).to(dev), [T(g["skip0"]).to(dev), T(g["skip1"]).to(dev), T(g["skip2"]).to(dev)]), g["out"], 5e-5)


def test_posenc(dev, golden):
    from cineflow.nn import position_embedding_sine_2d
    pos = position_embedding_sine_2d(8, 8, 32, dev)
    check(pos.view(1, 32, 8, 8), golden("posenc")["pos"], 1e-6)


def test_transformers(dev, golden):
    from cineflow.nn import CrossAttentionLayer, TransformerFlowEncoderSuccessiveNoEmb
    g = golden("crossattn")
    m = load(CrossAttentionLayer(32, 4, 1, 64), 8, dev)
    check(m(T(g["q"]).to(dev), T(g["k"]).to(dev), T(g["v"]).to(dev)), g["out"], 5e-5)
    g = golden("succ_transformer")
    m = load(TransformerFlowEncoderSuccessiveNoEmb(64, 8, 1), 9, dev)
    check(m(T(g["u"]).to(dev)), g["out"], 5e-5)


def test_generic_unet_and_tta(dev, golden):
    from cineflow.models import Generic_UNet
    from cineflow.inference import mirror_and_predict_2d
    g = golden("generic_unet")
    m = load(Generic_UNet(1, 8, 4, 3), 10, dev)
    check(m(T(g["x"]).to(dev)), g["logits"], 1e-4)
    g = golden("tta")
    check(mirror_and_predict_2d(m, T(g["x"]).to(dev), (0, 1), True, None), g["probs"], 2e-5)


def test_generic_unet_anisotropic_pooling(dev, golden):
    """pool_op_kernel_sizes (2,2), (2,2), (2,1) against the reference's output: the (2,1) stage runs its strided convolution at stride 1 +
    row subsampling and its transposed convolution as a 1x1 convolution + row interleave"""
    from cineflow.models import Generic_UNet
    g = golden("generic_unet_aniso")
    m = load(Generic_UNet(1, 8, 4, 3, pool_op_kernel_sizes=[[2, 2], [2, 2], [2, 1]]), 14, dev)
    check(m(T(g["x"]).to(dev)), g["logits"], 1e-4)


def test_generic_unet_3d_tta_and_tiled(dev, golden):
    """3-D rows of SURVEY 8a (a3 `_internal_predict_3D_3Dconv_tiled`, a4 8-flip TTA, a5 with conv_op = Conv3d) against
    the REFERENCE's outputs: anisotropic first stage ((1,3,3) kernels, (1,2,2) pooling), then isotropic."""
    from cineflow.models import Generic_UNet3D
    from cineflow.inference import mirror_and_predict_3d, predict_3D_3Dconv_tiled, _gaussian_on
    from oracle import ops as OO
    g = golden("generic_unet_3d")
    pool3, kern3 = [[1, 2, 2], [2, 2, 2]], [[1, 3, 3], [3, 3, 3], [3, 3, 3]]
    m = load(Generic_UNet3D(1, 4, 3, 2, pool_op_kernel_sizes=pool3, conv_kernel_sizes=kern3), 20, dev)
    x = T(g["x"]).to(dev)
    check(m(x), g["logits"], 1e-4, "Generic_UNet 3-D logits")
    check(mirror_and_predict_3d(m, x, (0, 1, 2), True, _gaussian_on(dev, (8, 16, 16))), g["tta"], 2e-5, "8-flip TTA x Gaussian")
    check(mirror_and_predict_3d(m, x, (1, 2), True, None), g["tta12"], 2e-5, "TTA axes (1,2)")
    seg, prob = predict_3D_3Dconv_tiled(m, g["vol"], (8, 16, 16), 0.5, True, (0, 1, 2), True, "constant", {"constant_values": 0})
    check(torch.from_numpy(prob), g["tiled_prob"], 2e-5, "tiled softmax")
    agree = float((seg == g["tiled_seg"]).mean())
    assert agree >= 0.999, agree
    for k in (1, 2):
        assert abs(OO.dice(seg, g["tiled_seg"], k) - 1.0) <= 1e-3
    # exact fp32 kernels give the same answer
    from cineflow import ops
    ops.set_conv_mode("f32")
    try:
        check(m(x), g["logits"], 1e-4, "fp32 mode")
    finally:
        ops.set_conv_mode("f16s")


@pytest.mark.parametrize("tag,ma,ff", [("ma", True, 64), ("cv", False, 48)])
def test_segflow_reference_golden(dev, golden, tag, ma, ff):
    """The reduced-width SegFlowGaussian against the REFERENCE's output (T=4): the north-star EPE bar."""
    from cineflow.models import SegFlowGaussian
    from oracle import ops as OO
    g = golden("segflow_" + tag)
    m = load(SegFlowGaussian(image_size=S, d_model=32, bottleneck_heads=4, dim_feedforward=ff, motion_appearance=ma, **RED), 11, dev)
    out = m(T(g["frames"]).to(dev))["backward_flow"].cpu()
    epe = OO.mean_epe(out, g["backward_flow"])
    assert epe <= 1e-4, "mean EPE %.3e px" % epe
    check(out, g["backward_flow"], 5e-4)


def test_segflow_reference_golden_exact_fp32_mode(dev, golden):
    """Same model with every convolution on the exact fp32 MFMA kernel (cineflow.ops.set_conv_mode('f32'))."""
    from cineflow import ops
    from cineflow.models import SegFlowGaussian
    from oracle import ops as OO
    g = golden("segflow_cv")
    m = load(SegFlowGaussian(image_size=S, d_model=32, bottleneck_heads=4, dim_feedforward=48, motion_appearance=False, **RED), 11, dev)
    ops.set_conv_mode("f32")
    try:
        out = m(T(g["frames"]).to(dev))["backward_flow"].cpu()
    finally:
        ops.set_conv_mode("f16s")
    assert OO.mean_epe(out, g["backward_flow"]) <= 1e-4


def test_successive_reference_golden(dev, golden):
    from cineflow.models import OpticalFlowModelSuccessive, ModelWrap
    from oracle import ops as OO
    g = golden("successive")
    m = load(ModelWrap(OpticalFlowModelSuccessive(S, 1, **RED), OpticalFlowModelSuccessive(S, 6, **RED)), 12, dev)
    o1, o2 = m(T(g["frames"]).to(dev))
    assert OO.mean_epe(o1["flow"].cpu(), g["flow1"]) <= 1e-4
    assert OO.mean_epe(o2["cumulated"].cpu(), g["cumulated"]) <= 1e-4
    gi = golden("successive_infer")
    out = m.model1(T(gi["frames"]).to(dev), inference=True)["flow"].cpu()
    assert OO.mean_epe(out, gi["flow"]) <= 1e-4


def test_raft_loop_vs_oracle(dev):
    """RAFT variant (parity unpinned: update block / CorrBlock restated from the published definition)."""
    from cineflow.models import SegFlowGaussian
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    # 128x128 so that the 4-level pyramid of the 16x16 feature map ends at 2x2 (a 1x1 level divides by zero in RAFT's sampler)
    kw = dict(image_size=128, in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=256, bottleneck_heads=4, dim_feedforward=64,
              motion_appearance=False, raft=True, raft_iters=3)
    m = load(SegFlowGaussian(**kw), 13, dev)
    ora = fill_module_(OM.SegFlowGaussian(**kw), 13)
    frames = randn(3, 1, 1, 128, 128, seed=38)
    out = m(frames.to(dev))["backward_flow"].cpu()
    with torch.no_grad():
        ref = ora(frames)["backward_flow"]
    assert out.shape == ref.shape == (3, 2, 1, 2, 128, 128)
    epe = OO.mean_epe(out, ref)
    assert epe <= 1e-4, "mean EPE %.3e px (|flow| mean %.3f)" % (epe, float(ref.abs().mean()))


def test_raft_net_reloaded_with_other_weights_matches_a_fresh_net(dev):
    """SepConvGRU's fused [r | z] gate weights and BasicUpdateBlock's scaled mask bias are derived on first use; a second
    load_state_dict on the same net (fold loop of predict_non_flow, user code) must drop them -- same numbers as a net built fresh."""
    from cineflow.models import SegFlowGaussian
    kw = dict(image_size=128, in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=256, bottleneck_heads=4, dim_feedforward=64,
              motion_appearance=False, raft=True, raft_iters=2)
    frames = randn(2, 1, 1, 128, 128, seed=39).to(dev)
    m = load(SegFlowGaussian(**kw), 13, dev)
    first = m(frames)["backward_flow"].clone()
    load(m, 14, dev)                                             # other weights into the SAME object
    again = m(frames)["backward_flow"]
    fresh = load(SegFlowGaussian(**kw), 14, dev)(frames)["backward_flow"]
    # (not bit-identical: the fused GroupNorm statistics meet in another order from run to run -- 5e-6 measured)
    assert float((again - fresh).abs().max()) <= 1e-4
    assert float((again - first).abs().max()) > 1e-2


def test_full_width_blocks_vs_oracle(dev):
    """Full-width (raft_config.yaml dims) single blocks at 256x256 against the oracle on the same seeded weights."""
    from cineflow.nn import ConvBlocks2DGroupLegacy, CrossAttentionLayer, ConvGRUCell
    from cineflow.weights import fill_module_
    from oracle import models as OM
    with torch.no_grad():
        x = randn(1, 6, 256, 256, seed=90)
        m = load(ConvBlocks2DGroupLegacy(6, 64, 1, residual=True), 20, dev)
        o = fill_module_(OM.ConvBlocks2DGroupLegacy(6, 64, 1, residual=True), 20)
        check(m(x.to(dev)), o(x), 5e-5, "DoubleConv 6->64 @256")
        x = randn(1, 128, 128, 128, seed=91)
        m = load(ConvBlocks2DGroupLegacy(128, 256, 1, residual=True, stride=2), 21, dev)
        o = fill_module_(OM.ConvBlocks2DGroupLegacy(128, 256, 1, residual=True, stride=2), 21)
        check(m(x.to(dev)), o(x), 5e-5, "DoubleConv 128->256 s2")
        q, k, v = randn(1, 256, 32, 32, seed=92), randn(1, 256, 32, 32, seed=93), randn(1, 256, 32, 32, seed=94)
        m = load(CrossAttentionLayer(256, 4, 1, 3072), 22, dev)
        o = fill_module_(OM.CrossAttentionLayer(256, 4, 1, 3072), 22)
        check(m(q.to(dev), k.to(dev), v.to(dev)), o(q, k, v), 5e-5, "CrossAttentionLayer 256")
        xg, h = randn(2, 256, 32, 32, seed=95), randn(2, 256, 32, 32, seed=96)
        m = load(ConvGRUCell((32, 32), 256, 256), 23, dev)
        o = fill_module_(OM.ConvGRUCell((32, 32), 256, 256), 23)
        check(m(xg.to(dev), h.to(dev)), o(xg, h), 5e-5, "ConvGRU 256")


def test_full_width_segflow_two_frames_vs_oracle(dev):
    """The full raft_config.yaml model (25.3 M parameters), one recurrence step at 256x256, against the CPU oracle."""
    from cineflow.models import SegFlowGaussian
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    for ma, ff in ((False, 2048), (True, 3072)):
        m = load(SegFlowGaussian(image_size=256, motion_appearance=ma, dim_feedforward=ff), 30, dev)
        ora = fill_module_(OM.SegFlowGaussian(image_size=256, motion_appearance=ma, dim_feedforward=ff), 30)
        frames = randn(3, 1, 1, 256, 256, seed=97)
        out = m(frames.to(dev))["backward_flow"].cpu()
        with torch.no_grad():
            ref = ora(frames)["backward_flow"]
        epe = OO.mean_epe(out, ref)
        assert epe <= 1e-4, "motion_appearance=%s: mean EPE %.3e px (|flow| mean %.3f)" % (ma, epe, float(ref.abs().mean()))


def test_sliding_window_segmentation_vs_oracle(dev):
    """BASELINE config 1 shapes scaled to test size: volume [1,3,70,60], patch (64,48) -> 2x2 tiles, Gaussian, 4 flips."""
    from cineflow.models import Generic_UNet
    from cineflow.inference import predict_3D_2Dconv_tiled
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    m = load(Generic_UNet(1, 8, 4, 3), 10, dev)
    ora = fill_module_(OM.GenericUNet2D(1, 8, 4, 3), 10)
    x = randn(1, 3, 70, 60, seed=98).numpy()
    seg, prob = predict_3D_2Dconv_tiled(m, x, (64, 48), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    with torch.no_grad():
        rseg, rprob = OM.predict_3d_2dconv_tiled(ora, x, (64, 48), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    assert seg.shape == rseg.shape == (3, 70, 60) and prob.shape == rprob.shape == (4, 3, 70, 60)
    assert float(np.abs(prob - rprob).max()) <= 5e-5
    for k in range(4):
        d = OO.dice(seg, rseg, k)
        assert np.isnan(d) or abs(d - 1.0) <= 1e-3
    # smaller-than-patch image (padding path) and single tile (no Gaussian)
    x = randn(1, 2, 40, 30, seed=99).numpy()
    seg, prob = predict_3D_2Dconv_tiled(m, x, (64, 48), do_mirroring=False)
    with torch.no_grad():
        rseg, rprob = OM.predict_3d_2dconv_tiled(ora, x, (64, 48), do_mirroring=False)
    assert float(np.abs(prob - rprob).max()) <= 5e-5


def test_baseline_config1_full_size_sliding_window_vs_oracle(dev):
    """BASELINE config 1 at its stated size: volume [1, 10, 256, 216], patch (256, 224) -> Y padded to 224, step 0.5,
    Gaussian weighting, 4 flips, Generic_UNet(32 base features, 6 pools, the plan's (2,1) last pooling) -- the whole 10-slice volume through
    predict_3D_2Dconv_tiled on the device, the oracle on slices 0, 4 and 9 (CPU: ~1 s per slice)."""
    from cineflow.models import Generic_UNet
    from cineflow.inference import predict_3D_2Dconv_tiled, compute_steps_for_sliding_window
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    pool = [[2, 2]] * 5 + [[2, 1]]      # nnU-Net's 2-D plan for this patch: 256 = 4 * 2^6 but 224 = 7 * 2^5 -> the sixth pooling halves the rows only
    m = load(Generic_UNet(1, 32, 4, 6, pool_op_kernel_sizes=pool), 41, dev)
    ora = fill_module_(OM.GenericUNet2D(1, 32, 4, 6, pool_op_kernel_sizes=pool), 41)
    x = smooth_cine(10, 1, 256, 7)[:, 0, :, :, :216].permute(1, 0, 2, 3).contiguous().numpy()        # [1, 10, 256, 216]
    assert x.shape == (1, 10, 256, 216)
    assert compute_steps_for_sliding_window((256, 224), (256, 224), 0.5) == [[0], [0]]                 # padded to the patch: one tile, no Gaussian (neural_network.py:657)
    seg, prob = predict_3D_2Dconv_tiled(m, x, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    assert seg.shape == (10, 256, 216) and prob.shape == (4, 10, 256, 216)
    for z in (0, 4, 9):
        with torch.no_grad():
            rseg, rprob = OM.predict_3d_2dconv_tiled(ora, x[:, z:z + 1], (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
        assert float(np.abs(prob[:, z] - rprob[:, 0]).max()) <= 5e-5, z
        for k in range(4):
            d = OO.dice(seg[z], rseg[0], k)
            assert np.isnan(d) or abs(d - 1.0) <= 1e-3
    # a larger field of view, so that the window really slides: [1, 2, 300, 260] -> 2 x 2 tiles of (256, 224), Gaussian importance map
    x2 = smooth_cine(2, 1, 320, 8)[:, 0, :, :300, :260].permute(1, 0, 2, 3).contiguous().numpy()
    seg2, prob2 = predict_3D_2Dconv_tiled(m, x2, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    with torch.no_grad():
        rseg2, rprob2 = OM.predict_3d_2dconv_tiled(ora, x2[:, :1], (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    assert float(np.abs(prob2[:, 0] - rprob2[:, 0]).max()) <= 5e-5
    assert float((seg2[0] == rseg2[0]).mean()) >= 0.9995


def test_model_wrap_successive_yaml_width_vs_oracle(dev):
    """BASELINE config 4's other dispatch at its real width (successive.yaml: in [6,128,256], out [64,128,256], d_model 512, 8 heads, FFN 2048,
    PatchMerging downsampling): ModelWrap(OpticalFlowModelSuccessive x 2), 256 x 256, T = 4 -- model1's adjacent flows and the ED -> t
    cumulative flows after two refinement steps of model2 against the CPU oracle (2 x 12.6 M parameters, same seeded weights)."""
    from cineflow.models import OpticalFlowModelSuccessive, ModelWrap
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    m = load(ModelWrap(OpticalFlowModelSuccessive(256, 1), OpticalFlowModelSuccessive(256, 6)), 17, dev)
    assert m.model1.d_model == 512
    ora = fill_module_(OM.ModelWrap(OM.OpticalFlowModelSuccessive(256, 1), OM.OpticalFlowModelSuccessive(256, 6)), 17)
    frames = smooth_cine(4, 1, 256, 23)
    o1, o2 = m(frames.to(dev))
    with torch.no_grad():
        r1, r2 = ora(frames)
    assert o1["flow"].shape == r1["flow"].shape == (3, 1, 2, 256, 256) and o2["cumulated"].shape == r2["cumulated"].shape == (3, 1, 2, 256, 256)
    e1 = OO.mean_epe(o1["flow"].cpu(), r1["flow"])
    e2 = OO.mean_epe(o2["cumulated"].cpu(), r2["cumulated"])
    assert e1 <= 1e-4 and e2 <= 1e-4, "mean EPE model1 %.3e, cumulated %.3e px (|flow| mean %.3f)" % (e1, e2, float(r2["cumulated"].abs().mean()))
    assert float(r2["cumulated"].abs().mean()) > 1e-3
    # the scaling-and-squaring branch of inference=True (Optical_flow_model_successive.py:399-402) at this width
    oi = m.model1(frames[:2].to(dev), inference=True)["flow"].cpu()
    with torch.no_grad():
        ri = ora.model1(frames[:2], inference=True)["flow"]
    assert OO.mean_epe(oi, ri) <= 1e-4


def test_joint_cine_pipeline_vs_oracle(dev):
    """BASELINE config 4 at test size: seg on every frame + two-chunk flow recurrence + label propagation."""
    from cineflow.models import SegFlowGaussian, Generic_UNet
    from cineflow.inference import predict_cine_slices, chunk_orders
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    kw = dict(image_size=64, in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=32, bottleneck_heads=4, dim_feedforward=48,
              motion_appearance=False)
    fnet = load(SegFlowGaussian(**kw), 11, dev)
    snet = load(Generic_UNet(1, 8, 4, 3), 10, dev)
    ofnet = fill_module_(OM.SegFlowGaussian(**kw), 11)
    osnet = fill_module_(OM.GenericUNet2D(1, 8, 4, 3), 10)
    Tn, B = 5, 2
    frames = randn(Tn, B, 1, 64, 64, seed=100)
    out = predict_cine_slices(fnet, snet, frames.to(dev))
    with torch.no_grad():
        probs = OM.mirror_and_predict_2d(osnet, frames.reshape(Tn * B, 1, 64, 64)).view(Tn, B, 4, 64, 64)
        seg = probs.argmax(2)
        flow = torch.zeros(Tn, B, 2, 64, 64)
        for order in chunk_orders(Tn):
            bf = ofnet(frames[order])["backward_flow"]
            for j, t in enumerate(order[1:]):
                flow[t] = bf[j]
        reg = OO.warp_labels(flow, seg[0][:, None].float())[:, :, 0]
    assert float((out["softmax"].cpu() - probs).abs().max()) <= 5e-5
    assert OO.mean_epe(out["flow"].cpu(), flow) <= 1e-4
    assert float((out["seg"].cpu().long() == seg).float().mean()) >= 0.9995
    for k in range(4):
        d = OO.dice(out["registered"].cpu().numpy(), reg.numpy(), k)
        assert np.isnan(d) or abs(d - 1.0) <= 1e-3


@pytest.mark.parametrize("ma", [False, True])
def test_joint_cine_pipeline_ragged_half_sequences(dev, ma):
    """T even (the bench's T = 30): the two ED-anchored half sequences differ in length by one; their common steps run as one batch of 2B
    sequences and the last step of the longer one alone (SegFlowGaussian.forward keep_from / keep).  Per-sequence flows must agree with the
    one-after-the-other schedule (no kernel mixes batch entries) and stay within the oracle bar."""
    from cineflow.models import SegFlowGaussian, Generic_UNet
    from cineflow import inference
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    kw = dict(image_size=64, in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32], d_model=32, bottleneck_heads=4, dim_feedforward=48,
              motion_appearance=ma)
    fnet = load(SegFlowGaussian(**kw), 21, dev)
    snet = load(Generic_UNet(1, 8, 4, 3), 20, dev)
    Tn, B = 6, 3
    o1, o2 = inference.chunk_orders(Tn)
    assert len(o1) == len(o2) + 1
    frames = randn(Tn, B, 1, 64, 64, seed=101)
    assert inference.RAGGED_CHUNKS
    out = inference.predict_cine_slices(fnet, snet, frames.to(dev))
    inference.RAGGED_CHUNKS = False
    try:
        seq = inference.predict_cine_slices(fnet, snet, frames.to(dev))
    finally:
        inference.RAGGED_CHUNKS = True
    # not bit-identical: the batch size picks the convolution shapes (tile sizes, fp32 summation order) and the order of the statistics
    # atomics; the schedules agree to ~5e-6 px, a twentieth of the bar
    assert float((out["flow"] - seq["flow"]).abs().max()) <= 2e-5
    assert float((out["registered"] == seq["registered"]).float().mean()) >= 0.9999
    ofnet = fill_module_(OM.SegFlowGaussian(**kw), 21)
    with torch.no_grad():
        flow = torch.zeros(Tn, B, 2, 64, 64)
        for order in (o1, o2):
            bf = ofnet(frames[order])["backward_flow"]
            for j, t in enumerate(order[1:]):
                flow[t] = bf[j]
    assert OO.mean_epe(out["flow"].cpu(), flow) <= 1e-4


# ------------------------------------------------------------------------------------------------ full size / long recurrence
def smooth_cine(T_, B, S_, seed):
    """z-scored synthetic cine frames of the bench (annulus + blobs + noise): realistic flows, not white noise"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.synthetic_cine(B, T_, S_, seed)


def drift_curve(out, ref):
    """mean EPE per recurrence step: out / ref [T-1,B,2,H,W]"""
    return [float(torch.sqrt(((out[t].cpu().double() - ref[t].double()) ** 2).sum(1)).mean()) for t in range(out.shape[0])]


def test_raft_full_size_12_iterations_vs_oracle(dev):
    """BASELINE config 3 at full size: SegFlowGaussian(raft=True, raft_iters=12), full width, 256x256 (feature maps
    [B,256,32,32], 4-level all-pairs pyramid) against the CPU oracle.  Parity unpinned for the update block / CorrBlock themselves
    (published RAFT, source absent from the reference).  The bar is the north star's 1e-4 px mean EPE, asserted for EVERY iteration's
    up-sampled flow of the frame pair.  A second pair is run in sequence (24 update steps on one GRU state, flows of ~9 px): RAFT's
    lookup differentiates a rough correlation surface, so any fp32 rounding difference is amplified ~1.2x per iteration; the same
    model with every convolution on the EXACT fp32 MFMA kernel is run beside it to show that this drift is fp32's own, not the
    f16 hi/lo split's."""
    from cineflow import ops
    from cineflow.models import SegFlowGaussian
    from cineflow.weights import fill_module_
    from oracle import models as OM
    kw = dict(image_size=256, motion_appearance=False, dim_feedforward=2048, raft=True, raft_iters=12)
    m = load(SegFlowGaussian(**kw), 31, dev)
    ora = fill_module_(OM.SegFlowGaussian(**kw), 31)
    frames = smooth_cine(3, 2, 256, 5)
    with torch.no_grad():
        ref = ora(frames)["backward_flow"]
    assert ref.shape == (12, 2, 2, 2, 256, 256)
    assert float(ref[-1].abs().mean()) > 1.0, "degenerate test: the flow should be pixels, not zeros"
    curves = {}
    for mode in ("f16s", "f32"):
        ops.set_conv_mode(mode)
        try:
            out = m(frames.to(dev))["backward_flow"].cpu()
        finally:
            ops.set_conv_mode("f16s")
        assert out.shape == ref.shape
        curves[mode] = [drift_curve(out[:, t], ref[:, t]) for t in range(2)]
        print("RAFT full size, %s convolutions: mean EPE per iteration, pair 1: %s | pair 2: %s" %
              (mode, " ".join("%.1e" % e for e in curves[mode][0]), " ".join("%.1e" % e for e in curves[mode][1])))
    assert max(curves["f16s"][0]) <= 1e-4, "frame pair, 12 iterations: mean EPE %.3e px" % max(curves["f16s"][0])
    assert max(curves["f32"][0]) <= 1e-4
    # second pair in sequence: amplified fp32 rounding noise; the f16-split path must not drift more than the exact-fp32 path does
    assert max(curves["f16s"][1]) <= max(3e-4, 2.0 * max(curves["f32"][1])), (max(curves["f16s"][1]), max(curves["f32"][1]))


def test_long_recurrence_reduced_width_T30(dev):
    """30-frame recurrence (cumulated += flow, ConvGRU state, memory encoder fed with the warped error) at reduced width: the per-step
    drift of the f16-split path against the fp32 CPU oracle stays under the 1e-4 px bar at every step, for both dispatches."""
    from cineflow.models import SegFlowGaussian
    from cineflow.weights import fill_module_
    from oracle import models as OM
    for ma, ff in ((False, 48), (True, 64)):
        kw = dict(image_size=S, d_model=32, bottleneck_heads=4, dim_feedforward=ff, motion_appearance=ma, **RED)
        m = load(SegFlowGaussian(**kw), 11, dev)
        ora = fill_module_(OM.SegFlowGaussian(**kw), 11)
        frames = smooth_cine(30, 2, S, 7)
        out = m(frames.to(dev))["backward_flow"]
        with torch.no_grad():
            ref = ora(frames)["backward_flow"]
        curve = drift_curve(out, ref)
        print("T=30 reduced width (motion_appearance=%s): EPE step 1 %.1e, 10 %.1e, 20 %.1e, 29 %.1e; |flow| %.2f px"
              % (ma, curve[0], curve[9], curve[19], curve[28], float(ref[-1].abs().mean())))
        assert max(curve) <= 1e-4, "max per-step mean EPE %.3e at step %d" % (max(curve), int(np.argmax(curve)) + 1)


def test_long_recurrence_full_width_T16(dev):
    """The full 25 M-parameter video.yaml model over 16 frames at 256x256 -- 15 recurrence steps, the longest chain the pipeline runs on a
    30-frame cine (two ED-anchored half sequences of 16 and 15 frames); the CPU oracle takes ~20 s: per-step drift curve of the f16
    hi/lo-split convolutions (the dropped lo*lo term every layer) against the fp32 oracle."""
    from cineflow.models import SegFlowGaussian
    from cineflow.weights import fill_module_
    from oracle import models as OM
    kw = dict(image_size=256, motion_appearance=False, dim_feedforward=2048)
    m = load(SegFlowGaussian(**kw), 30, dev)
    ora = fill_module_(OM.SegFlowGaussian(**kw), 30)
    frames = smooth_cine(16, 1, 256, 9)
    out = m(frames.to(dev))["backward_flow"]
    with torch.no_grad():
        ref = ora(frames)["backward_flow"]
    curve = drift_curve(out, ref)
    print("T=16 full width: mean EPE per step " + " ".join("%.1e" % e for e in curve) + "; |flow| %.2f px" % float(ref[-1].abs().mean()))
    assert max(curve) <= 1e-4, "max per-step mean EPE %.3e" % max(curve)


def test_config4_full_size_one_call(dev):
    """BASELINE config 4 as ONE call at its stated size -- predict_cine_slices with the full-width video.yaml SegFlowGaussian
    (25 M parameters) + the 32-base / 6-pool Generic_UNet on 256 x 256, T = 30, B = 2 slices: the 16 / 15-frame ED-anchored half chains share
    launches (ragged schedule).  (i) ragged == one-after-the-other schedule to 2e-5 px; (ii) at the two chain ENDS (the frames with the longest
    recurrence behind them: 15 and 14 steps) slice 0's flow against the fp32 CPU oracle (mean EPE <= 1e-4 px) and the propagated labels
    (Dice within 1e-3); the oracle runs the two chains of slice 0 once (~1 min of CPU)."""
    from cineflow.models import SegFlowGaussian, Generic_UNet
    from cineflow import inference
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    kw = dict(image_size=256, motion_appearance=False, dim_feedforward=2048)
    fnet = load(SegFlowGaussian(**kw), 30, dev)
    snet = load(Generic_UNet(1, 32, 4, 6), 41, dev)
    Tn, B = 30, 2
    frames = smooth_cine(Tn, B, 256, 13)
    o1, o2 = inference.chunk_orders(Tn)
    assert len(o1) == 16 and len(o2) == 15
    assert inference.RAGGED_CHUNKS
    out = inference.predict_cine_slices(fnet, snet, frames.to(dev))
    inference.RAGGED_CHUNKS = False
    try:
        seq = inference.predict_cine_slices(fnet, snet, frames.to(dev))
    finally:
        inference.RAGGED_CHUNKS = True
    assert out["flow"].shape == (Tn, B, 2, 256, 256) and out["registered"].shape == (Tn, B, 256, 256)
    # not bit-identical: the batch size picks the convolution shapes (fp32 summation order) and the order of the statistics atomics, and 15
    # recurrence steps of a randomly initialised 25 M-parameter network amplify that noise where the flow field is steep: the schedules agree
    # to ~1e-6 px in the mean; the worst single pixel of the 7.9 M is printed
    dm = float(torch.sqrt(((out["flow"] - seq["flow"]) ** 2).sum(2)).mean())
    d = float((out["flow"] - seq["flow"]).abs().max())
    print("config 4 full size: ragged vs sequential schedule: mean EPE %.2e px, max |diff| %.2e px" % (dm, d))
    assert dm <= 2e-5 and d <= 2e-3, "ragged vs sequential schedule: mean EPE %.3e px, max |flow diff| %.3e px" % (dm, d)
    assert float((out["registered"] == seq["registered"]).float().mean()) >= 0.9999
    assert float((out["seg"] == seq["seg"]).float().mean()) >= 0.9999
    # ---- oracle on slice 0: both chains once, the ED segmentation, the label warp at the chain ends
    ofnet = fill_module_(OM.SegFlowGaussian(**kw), 30)
    osnet = fill_module_(OM.GenericUNet2D(1, 32, 4, 6), 41)
    ends = (o1[-1], o2[-1])
    with torch.no_grad():
        ref = {}
        for order in (o1, o2):
            bf = ofnet(frames[order][:, :1])["backward_flow"]
            ref[order[-1]] = bf[-1]                                            # [1,2,256,256]: ED -> chain end
        ed_seg = OM.mirror_and_predict_2d(osnet, frames[0, :1]).argmax(1)       # [1,256,256]
    for t in ends:
        epe = OO.mean_epe(out["flow"][t, :1].cpu(), ref[t])
        assert epe <= 1e-4, "chain end frame %d: mean EPE %.3e px (|flow| mean %.3f px)" % (t, epe, float(ref[t].abs().mean()))
        reg = OO.warp_labels(ref[t][None], ed_seg[:, None].float())[0, :, 0]     # [1,256,256]
        got = out["registered"][t, :1].cpu().numpy()
        for k in range(4):
            dk = OO.dice(got, reg.numpy(), k)
            assert np.isnan(dk) or abs(dk - 1.0) <= 1e-3, "chain end frame %d class %d: Dice %.5f" % (t, k, dk)
    assert torch.equal(out["seg"][0, :1].cpu().long(), ed_seg) or float((out["seg"][0, :1].cpu().long() == ed_seg).float().mean()) >= 0.9995


def test_generic_unet_bench_width_vs_oracle(dev):
    """The network the bench runs -- Generic_UNet(1, 32, 4, 6): 32 base features, 6 pools, 480-channel 4x4 bottleneck -- on two
    256x256 frames against the oracle (which golden `generic_unet.npz` ties to the reference at reduced width)."""
    from cineflow.models import Generic_UNet
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    m = load(Generic_UNet(1, 32, 4, 6), 41, dev)
    ora = fill_module_(OM.GenericUNet2D(1, 32, 4, 6), 41)
    x = smooth_cine(2, 1, 256, 3).reshape(2, 1, 256, 256)
    out = m(x.to(dev)).cpu()
    with torch.no_grad():
        ref = ora(x)
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 5e-5 * max(1.0, scale), "logits max|diff| %.3e (scale %.2f)" % (float((out - ref).abs().max()), scale)
    a, b = out.argmax(1).numpy(), ref.argmax(1).numpy()
    for k in range(4):
        d = OO.dice(a, b, k)
        assert np.isnan(d) or abs(d - 1.0) <= 1e-3


def test_generic_unet_mixed_precision_measured(dev):
    """`mixed_precision=True` on the segmentation path = the U-Net's convolutions in the one-term product mode
    (ops.conv_terms(1): operands rounded to fp16, fp32 accumulation and norms -- the reference's fp16 autocast, neural_network.py:140-146).
    RESULT (stated either way, as asked): on the seeded random weights every test and the bench use, the north-star Dice bar (1e-3) does NOT
    hold -- the softmax moves by ~2e-3 and a random network's logits are near-ties, so ~0.15 % of the voxels change class (per-class Dice
    0.992-0.999 at the bench width on 256 x 256; measured again here).  The mode therefore stays opt-in (CF_SEG_MIXED_PRECISION=1 /
    bench.py --seg-precision f16) and the default path ignores the flag and stays f32-class.  This test pins what the mode does: softmax
    within 2e-2 of the fp32 oracle, per-class Dice >= 0.99, through the plain forward and through BASELINE config 1's sliding window; and that
    leaving the block restores the three-term mode (5e-5)."""
    from cineflow import ops
    from cineflow.inference import mirror_and_predict_2d, predict_3D_2Dconv_tiled
    from cineflow.models import Generic_UNet
    from cineflow.weights import fill_module_
    from oracle import models as OM
    from oracle import ops as OO
    m = load(Generic_UNet(1, 32, 4, 6), 41, dev)
    ora = fill_module_(OM.GenericUNet2D(1, 32, 4, 6), 41)
    x = smooth_cine(3, 1, 256, 3).reshape(3, 1, 256, 256)
    with ops.conv_terms(1):
        p16 = mirror_and_predict_2d(m, x.to(dev)).cpu()
    p32 = mirror_and_predict_2d(m, x.to(dev)).cpu()
    with torch.no_grad():
        pref = OM.mirror_and_predict_2d(ora, x)
    a, b = p16.argmax(1).numpy(), pref.argmax(1).numpy()
    dice = [OO.dice(a, b, k) for k in range(4)]
    print("one-term U-Net, 256x256: max |softmax - oracle| %.2e (three-term path: %.2e); voxels differing from the oracle's arg-max: %d of %d; Dice %s"
          % (float((p16 - pref).abs().max()), float((p32 - pref).abs().max()), int((a != b).sum()), a.size, " ".join("%.5f" % d for d in dice)))
    assert float((p16 - pref).abs().max()) <= 2e-2 and float((p32 - pref).abs().max()) <= 5e-5
    assert all(np.isnan(d) or d >= 0.99 for d in dice)
    # BASELINE config 1: sliding window + Gaussian + 4 flips with the plans' anisotropic pooling, 2 slices of the stated volume
    pools = [[2, 2]] * 5 + [[2, 1]]
    m1 = load(Generic_UNet(1, 32, 4, 6, pool_op_kernel_sizes=pools), 43, dev)
    o1 = fill_module_(OM.GenericUNet2D(1, 32, 4, 6, pool_op_kernel_sizes=pools), 43)
    vol = smooth_cine(2, 1, 256, 7)[:, 0, :, :, :216].permute(1, 0, 2, 3).contiguous().numpy()        # [1, 2, 256, 216]
    with ops.conv_terms(1):
        seg16, _ = predict_3D_2Dconv_tiled(m1, vol, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    with torch.no_grad():
        seg_ref, _ = OM.predict_3d_2dconv_tiled(o1, vol, (256, 224), step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True)
    dice1 = [OO.dice(np.asarray(seg16), np.asarray(seg_ref), k) for k in range(4)]
    print("one-term U-Net, config 1 sliding window: Dice %s" % " ".join("%.5f" % d for d in dice1))
    assert all(np.isnan(d) or d >= 0.99 for d in dice1)


# ------------------------------------------------------------------------------------------------ Processor centroid path
def test_processor_centroid_vs_oracle(dev):
    """Processor.discretize / get_mean_centroid / preprocess_no_registration (processor.py:140-176, 232-237) with a 2-class network:
    device path (batched frames, cf_frame_boxes) against the oracle's frame-by-frame restatement, empty frames included."""
    from cineflow.inference import Processor, CroppingNet
    from cineflow.models import Generic_UNet
    from cineflow.weights import fill_module_
    from oracle import models as OM
    net = load(Generic_UNet(1, 8, 2, 3), 51, dev)
    onet = fill_module_(OM.GenericUNet2D(1, 8, 2, 3), 51)
    proc = Processor(32, 64, CroppingNet(net))
    oproc = OM.Processor(32, 64, lambda x: {"pred": onet(x)})
    frames = smooth_cine(6, 1, 64, 21)[:, 0] * 40 + 90          # [T,1,64,64]
    frames[2] = 0                                               # an all-zero frame: no network, empty mask
    frames[4] = 7.0                                             # a constant non-zero frame: std 0 -> only the mean is subtracted (zeros into the network, no NaN)
    cen, lab = proc.preprocess_no_registration(frames.to(dev))
    from oracle import ops as OO
    with torch.no_grad():
        ocen, olab = oproc.preprocess_no_registration(frames)
        ologit = torch.stack([onet(OO.normalize_intensity(frames[t][None]))[0] for t in range(6) if t != 2])
    # random weights leave the two logits nearly tied over smooth regions: the label maps must agree wherever the oracle's margin is
    # above the 5e-5 logit tolerance, and the frame without signal is empty
    sure = (ologit[:, 0] - ologit[:, 1]).abs() > 1e-3
    keep = [t for t in range(6) if t != 2]
    assert float(sure.float().mean()) > 0.5
    assert bool((lab.cpu().long()[keep][sure] == olab[keep][sure]).all()) and int(lab[2].max()) == 0 and int(olab[2].max()) == 0
    assert bool(sure[keep.index(4)].any()), "the constant frame must be compared somewhere"
    if float((lab.cpu().long() == olab).float().mean()) == 1.0:
        assert cen.tolist() == ocen.tolist()
    assert proc.get_mean_centroid(olab.to(torch.uint8).to(dev)).tolist() == ocen.tolist()
    # the box arithmetic on hand-made masks, one empty
    m = torch.zeros(4, 40, 56, dtype=torch.uint8)
    m[0, 5:11, 7:30] = 1
    m[1, 39, 55] = 1
    m[3, 0:40, 20:21] = 1
    assert proc.get_mean_centroid(m.to(dev)).tolist() == oproc.get_mean_centroid(m.long()).tolist()
    from cineflow import ops
    assert ops.frame_boxes(m.to(dev)).cpu().tolist() == [[7, 5, 29, 10], [55, 39, 55, 39], [-1, -1, -1, -1], [20, 0, 20, 39]]
    assert ops.frame_boxes(m.float().to(dev)).cpu().tolist() == ops.frame_boxes(m.to(dev)).cpu().tolist()

