"""CPU tests: the oracle (oracle/) against the golden vectors produced by the reference classes
(tests/golden/make_golden.py) and against the reference's own known-answer test.  No GPU, no /root/reference."""
import numpy as np
import pytest
import torch

from cineflow.weights import fill_module_
from oracle import models as OM
from oracle import ops as OO

RED = dict(in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32])
S = 64


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    d = float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())
    assert d <= tol, "max|diff| %.3e > %.1e" % (d, tol)


# ---------------------------------------------------------------- reference known-answer test (a1)
def test_steps_known_answers_of_the_reference_test_file():
    """tests/test_steps_for_sliding_window_prediction.py:96-163 of the reference, answer for answer."""
    f = OO.compute_steps_for_sliding_window
    assert f((64, 130), (128, 260), 0.5) == [[0, 32, 64], [0, 65, 130]]
    assert f((64, 130), (128, 260), 0.85) == [[0, 32, 64], [0, 65, 130]]
    assert f((64, 130), (128, 260), 1) == [[0, 64], [0, 130]]
    assert f((128, 128, 128), (146, 176, 148), 0.5) == [[0, 18], [0, 48], [0, 20]]
    assert f((80, 192, 160), (130, 320, 244), 0.5) == [[0, 25, 50], [0, 64, 128], [0, 42, 84]]
    assert f((80, 192, 160), (130, 320, 244), 0.75) == [[0, 50], [0, 128], [0, 84]]
    assert f((128, 128, 128), (424, 456, 456), 0.5) == [[0, 59, 118, 178, 237, 296], [0, 55, 109, 164, 219, 273, 328],
                                                        [0, 55, 109, 164, 219, 273, 328]]
    assert f((40, 56, 40), (40, 56, 40), 0.5) == [[0], [0], [0]]
    assert f((64, 192, 192), (94, 308, 308), 0.5) == [[0, 30], [0, 58, 116], [0, 58, 116]]
    for step in (1, 0.125, 0.5):
        assert f((24, 845, 321), (24, 845, 321), step) == [[0], [0], [0]]
        assert f((123, 143), (123, 143), step) == [[0], [0]]


def test_steps_random_invariants():
    """the four invariants of the reference test (:39-58) on 2000 random cases."""
    rng = np.random.RandomState(0)
    for _ in range(2000):
        dim = rng.choice((2, 3))
        patch = tuple(int(v) for v in rng.randint(16, 1024, dim))
        image = tuple(max(int(rng.randint(p // 2, p * 10)), p) for p in patch)
        step = float(rng.uniform(0.01, 1))
        steps = OO.compute_steps_for_sliding_window(patch, image, step)
        target = [i * step for i in patch]
        for d in range(dim):
            assert steps[d][0] == 0
            assert steps[d][-1] + patch[d] == image[d]
            assert all(steps[d][i + 1] <= steps[d][i] + patch[d] for i in range(len(steps[d]) - 1))
            assert all(steps[d][i] + np.ceil(target[d]) >= steps[d][i + 1] for i in range(len(steps[d]) - 1))


# ---------------------------------------------------------------- golden vectors
def test_gaussian(golden):
    g = golden("gaussian")
    close(OO.get_gaussian((64, 64)), g["g64"], 0)
    g256 = OO.get_gaussian((256, 224))
    close(g256[:16, :16], g["g256_corner"], 0)
    close(g256[120:136, 104:120], g["g256_center"], 0)
    assert abs(float(g256.astype(np.float64).sum()) - float(g["g256_sum"])) < 1e-9
    assert g256.max() == 1.0 and g256.min() > 0


@pytest.mark.parametrize("tag", ["32", "40x24", "3d"])
def test_warp_and_vecint(golden, tag):
    g = golden("warp_" + tag)
    close(OO.warp_bilinear(T(g["flow"]).clone(), T(g["src"])), g["warped"], 0)
    close(OO.vecint(T(g["flow"]).clone()), g["vecint"], 0)


def test_warp_256(golden):
    g = golden("warp_256")
    gen = torch.Generator().manual_seed(13)
    src = torch.randn(1, 4, 256, 256, generator=gen)
    out = OO.warp_bilinear(T(g["flow"]).float(), src)
    # flow was stored as fp16: regenerate exactly instead
    gen = torch.Generator().manual_seed(12)
    flow = torch.nn.functional.avg_pool2d(torch.randn(1, 2, 288, 288, generator=gen), 33, stride=1) * 120.0
    out = OO.warp_bilinear(flow, src)
    close(out[:, :, :16, :16], g["warped_corner"], 0)
    close(out[:, :, 120:136, 120:136], g["warped_center"], 0)
    assert abs(float(out.double().sum()) - float(g["checksum"])) < 1e-6
    assert abs(float(out.double().abs().sum()) - float(g["abs_checksum"])) < 1e-6


def test_warp_labels(golden):
    g = golden("warp_labels")
    out = OO.warp_labels(T(g["flow"]), T(g["labels"]))
    assert torch.equal(out.to(torch.int8), T(g["registered"]))


def test_jacobian_matches_numpy_definition(golden):
    g = golden("jacobian")
    close(OO.jacobian_determinant(g["disp"]), g["det"], 0)
    close(OO.jacobian_determinant(g["disp3"]), g["det3"], 0)
    # identity displacement -> determinant 1 everywhere
    close(OO.jacobian_determinant(np.zeros((8, 9, 2))), np.ones((8, 9)), 0)


def test_posenc(golden):
    close(OM.position_embedding_sine_2d(1, 8, 8, 16), golden("posenc")["pos"], 0)


def test_convgru(golden):
    g = golden("convgru")
    m = fill_module_(OM.ConvGRUCell((8, 8), 32, 32), 1)
    with torch.no_grad():
        close(m(T(g["x"]), T(g["h"])), g["out"], 1e-6)


@pytest.mark.parametrize("tag,kw", [("res_s1", dict(in_dim=6, out_dim=16, nb_blocks=1, residual=True)),
                                    ("res_s2", dict(in_dim=16, out_dim=32, nb_blocks=1, residual=True, stride=2)),
                                    ("same", dict(in_dim=16, out_dim=16, nb_blocks=1, residual=True)),
                                    ("nores", dict(in_dim=16, out_dim=8, nb_blocks=1, residual=False)),
                                    ("single", dict(in_dim=8, out_dim=16, nb_blocks=1, residual=True, nb_conv=1))])
def test_convblocks(golden, tag, kw):
    g = golden("convblock_" + tag)
    m = fill_module_(OM.ConvBlocks2DGroupLegacy(**kw), 2)
    with torch.no_grad():
        close(m(T(g["x"])), g["out"], 2e-6)


def test_patch_expand_merge(golden):
    g = golden("patchexpand")
    with torch.no_grad():
        close(fill_module_(OM.PatchExpand2DGroup(32, 16), 3)(T(g["x"])), g["out"], 2e-6)
        g = golden("patchmerging")
        close(fill_module_(OM.PatchMerging2DGroup(8, 16), 3)(T(g["x"])), g["out"], 2e-6)


def test_encoders_decoder(golden):
    with torch.no_grad():
        g = golden("encoder2d")
        m = fill_module_(OM.Encoder2D(d_model=32, conv_depth=[1, 1, 1], in_dims=RED["in_dims"], out_dims=RED["out_encoder_dims"],
                                      nb_conv=2, extra_block=True, residual=True, downsample_conv=2), 4)
        f, sk = m(T(g["x"]))
        close(f, g["feat"], 5e-6)
        for i in range(3):
            close(sk[i], g["skip%d" % i], 5e-6)
        g = golden("encoder_ma")
        m = fill_module_(OM.Encoder2D(d_model=32, conv_depth=[1, 1, 1], in_dims=[2, 16, 32], out_dims=RED["out_encoder_dims"],
                                      nb_conv=2, extra_block=False, residual=True, downsample_conv=2, motion_appearance=True), 5)
        a, mo, sk = m(T(g["x"]))
        close(a, g["app"], 5e-6)
        close(mo, g["motion"], 5e-6)
        g = golden("encoder2d_succ")
        m = fill_module_(OM.Encoder2D(d_model=64, conv_depth=[1, 1, 1], in_dims=RED["in_dims"], out_dims=RED["out_encoder_dims"],
                                      nb_conv=2, extra_block=False, residual=False, downsample_conv=1), 6)
        f, sk = m(T(g["x"]))
        close(f, g["feat"], 5e-6)
        g = golden("decoder2d")
        m = fill_module_(OM.Decoder2D(d_model=32, dot_multiplier=2, conv_depth=[1, 1, 1], in_encoder_dims=[32, 16, 4],
                                      out_encoder_dims=[32, 16, 8], num_classes=2, nb_conv=2, residual=True), 7)
        close(m(T(g["x"]), [T(g["skip0"]), T(g["skip1"]), T(g["skip2"])]), g["out"], 5e-6)


def test_transformers(golden):
    with torch.no_grad():
        g = golden("crossattn")
        m = fill_module_(OM.CrossAttentionLayer(dim=32, nhead=4, num_layers=1, dim_feedforward=64), 8)
        close(m(T(g["q"]), T(g["k"]), T(g["v"])), g["out"], 5e-6)
        g = golden("succ_transformer")
        m = fill_module_(OM.TransformerFlowEncoderSuccessiveNoEmb(dim=64, nhead=8, num_layers=1), 9)
        close(m(T(g["u"])), g["out"], 5e-6)


def test_generic_unet_and_tta(golden):
    with torch.no_grad():
        g = golden("generic_unet")
        m = fill_module_(OM.GenericUNet2D(1, 8, 4, 3), 10)
        close(m(T(g["x"])), g["logits"], 1e-5)
        g = golden("tta")
        close(OM.mirror_and_predict_2d(m, T(g["x"]), (0, 1), True, None), g["probs"], 1e-6)


def test_generic_unet_anisotropic_pooling(golden):
    """the plans' per-stage pooling kernels (here (2,2), (2,2), (2,1): the ACDC 2-D plan ends with (2,1) for its (256, 224) patch) against the
    reference's Generic_UNet built with that pool_op_kernel_sizes"""
    with torch.no_grad():
        g = golden("generic_unet_aniso")
        m = fill_module_(OM.GenericUNet2D(1, 8, 4, 3, pool_op_kernel_sizes=[[2, 2], [2, 2], [2, 1]]), 14)
        close(m(T(g["x"])), g["logits"], 1e-5)


def test_connected_component_filter(golden):
    g = golden("connected_components")
    cases = [([1, 2, 3], None), ([(1, 2), 3], None), ([1, 2], {1: 40.0, 2: 1e9}), (None, None)]
    for i, (fw, mv) in enumerate(cases):
        out, _, kept = OO.remove_all_but_the_largest_connected_component(g["labels"].copy(), fw, 1.5, mv)
        assert (out == g["out%d" % i]).all()


def test_generic_unet_3d_tta_and_tiled(golden):
    """3-D rows (a3/a4/a5 with conv_op = Conv3d): the oracle against the reference's own Generic_UNet, 8-flip TTA and
    _internal_predict_3D_3Dconv_tiled outputs."""
    g = golden("generic_unet_3d")
    pool3, kern3 = [[1, 2, 2], [2, 2, 2]], [[1, 3, 3], [3, 3, 3], [3, 3, 3]]
    with torch.no_grad():
        m = fill_module_(OM.GenericUNet3D(1, 4, 3, 2, pool_op_kernel_sizes=pool3, conv_kernel_sizes=kern3), 20)
        close(m(T(g["x"])), g["logits"], 1e-5)
        g3 = torch.from_numpy(OO.get_gaussian((8, 16, 16)))
        close(OM.mirror_and_predict_3d(m, T(g["x"]), (0, 1, 2), True, g3), g["tta"], 1e-6)
        close(OM.mirror_and_predict_3d(m, T(g["x"]), (1, 2), True, None), g["tta12"], 1e-6)
        seg, prob = OM.predict_3d_tiled(m, g["vol"], (8, 16, 16), 0.5, True, (0, 1, 2), True, "constant", {"constant_values": 0})
    close(prob, g["tiled_prob"], 1e-6)
    assert (seg == g["tiled_seg"]).all()


@pytest.mark.parametrize("tag,ma,ff", [("ma", True, 64), ("cv", False, 48)])
def test_segflow(golden, tag, ma, ff):
    g = golden("segflow_" + tag)
    m = fill_module_(OM.SegFlowGaussian(image_size=S, d_model=32, bottleneck_heads=4, dim_feedforward=ff, motion_appearance=ma, **RED), 11)
    with torch.no_grad():
        out = m(T(g["frames"]))["backward_flow"]
    close(out, g["backward_flow"], 2e-5)
    assert OO.mean_epe(out, g["backward_flow"]) <= 1e-5


def test_successive(golden):
    g = golden("successive")
    m = OM.ModelWrap(OM.OpticalFlowModelSuccessive(S, 1, **RED), OM.OpticalFlowModelSuccessive(S, 6, **RED))
    fill_module_(m, 12)
    with torch.no_grad():
        o1, o2 = m(T(g["frames"]))
        close(o1["flow"], g["flow1"], 2e-5)
        close(o2["cumulated"], g["cumulated"], 2e-5)
        gi = golden("successive_infer")
        close(m.model1(T(gi["frames"]), inference=True)["flow"], gi["flow"], 2e-5)


# ---------------------------------------------------------------- unpinned pieces: internal consistency only
def test_corr_volume_definition():
    """CorrVolume is parity-unpinned (source absent).  Check the vectorised oracle against a literal loop."""
    g = torch.Generator().manual_seed(3)
    cur, prev = torch.randn(1, 5, 12, 10, generator=g), torch.randn(1, 5, 12, 10, generator=g)
    out = OO.corr_volume(cur, prev, radius=2, stride=2)
    for (dy, dx, y, x) in [(-2, -2, 0, 0), (0, 0, 5, 5), (2, 1, 7, 3), (-1, 2, 11, 9), (2, 2, 11, 9)]:
        yy, xx = y + dy * 2, x + dx * 2
        want = float((cur[0, :, y, x] * prev[0, :, yy, xx]).mean()) if 0 <= yy < 12 and 0 <= xx < 10 else 0.0
        assert abs(float(out[0, (dy + 2) * 5 + (dx + 2), y, x]) - want) < 1e-6


def test_corr_lookup_integer_coords_pick_volume_entries():
    g = torch.Generator().manual_seed(4)
    f1, f2 = torch.randn(1, 6, 8, 8, generator=g), torch.randn(1, 6, 8, 8, generator=g)
    vol = OO.corr_allpairs(f1, f2)
    pyr = OO.corr_pyramid(vol, 2)
    coords = OO.coords_grid(1, 8, 8)
    out = OO.corr_lookup(pyr, coords, radius=1)
    # centre tap (i=j=1) of level 0 at integer coords is the volume entry itself
    for (y, x) in [(0, 0), (3, 4), (7, 7)]:
        assert abs(float(out[0, 4, y, x]) - float(vol[0, y * 8 + x, y, x])) < 1e-5


def test_pad_nd_image():
    x = np.arange(2 * 5 * 7, dtype=np.float32).reshape(2, 5, 7)
    r, sl = OO.pad_nd_image(x, (8, 8), "constant", {"constant_values": 0}, True)
    assert r.shape == (2, 8, 8)
    assert np.array_equal(r[tuple(sl)], x)
    assert sl[1] == slice(1, 6) and sl[2] == slice(0, 7)
    r2 = OO.pad_nd_image(x, (4, 4))
    assert r2 is x


def test_processor_roundtrip():
    p = OM.Processor(crop_size=16, image_size=40)
    for c in [(20, 20), (2, 3), (39, 39), (8, 33)]:
        pl = p.adjust_cropping_window(c)
        x0, x1, y0, y1 = pl["crop_indices"]
        assert x1 - x0 == 16 and y1 - y0 == 16 and x0 >= 0 and y0 >= 0 and x1 <= 40 and y1 <= 40
        data = torch.arange(40 * 40, dtype=torch.float32).view(1, 1, 40, 40)
        crop, pad = p.crop_and_pad(data, c)
        back = p.uncrop_no_registration(crop[None], pad[None])[0]
        assert back.shape == data.shape
        assert torch.equal(back[..., y0:y1, x0:x1], data[..., y0:y1, x0:x1])
        assert float(back.sum()) == float(data[..., y0:y1, x0:x1].sum())


def test_dice_and_epe():
    a = np.array([[1, 1, 0], [0, 2, 2]])
    b = np.array([[1, 0, 0], [0, 2, 2]])
    assert abs(OO.dice(a, b, 1) - 2 / 3) < 1e-12
    assert OO.dice(a, b, 2) == 1.0
    assert np.isnan(OO.dice(a, b, 3))
    f = torch.zeros(1, 2, 4, 4)
    g = f.clone()
    g[:, 0] += 3
    g[:, 1] += 4
    assert abs(OO.mean_epe(f, g) - 5.0) < 1e-12


def test_processor_mean_centroid_known_answers():
    """Processor.get_mean_centroid (processor.py:140-160) worked by hand: bounding-box centres per frame, (H/2, W/2) -- in that order --
    for an empty frame, the mean truncated by .int()."""
    import torch
    from oracle import models as OM
    p = OM.Processor(32, 64)
    m = torch.zeros(3, 20, 30, dtype=torch.int64)
    m[0, 4:9, 10:21] = 2          # x 10..20, y 4..8  -> (15, 6)
    m[2, 0:2, 0:5] = 1            # x 0..4,  y 0..1  -> (2, 0.5)
    assert p.get_mean_centroid(m).tolist() == [9, 7]          # frame 1 is empty: (H/2, W/2) = (10, 15); means (9.0, 7.17)
    m[1, 19, 29] = 3              # a single pixel: (29, 19)
    assert p.get_mean_centroid(m).tolist() == [15, 8]         # (15+29+2)/3 = 15.33, (6+19+0.5)/3 = 8.5
    assert OM.masks_to_boxes(m[:1]).tolist() == [[10.0, 4.0, 20.0, 8.0]]
