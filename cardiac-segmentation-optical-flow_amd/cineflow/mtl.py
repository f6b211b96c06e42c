"""`MTLmodel` -- the fork's own segmenter and the Processor's cropping network -- on the HIP C ABI.

nnunet/network_architecture/MTL_model.py:84-470 as `adversarial_acdc.yaml` / `seg_model.yaml` configure it (no Swin stages in the
encoder: `transformer_depth: []`; BatchNorm; transformer bottleneck; asymmetric decoder; Swin cross-attention filter on every skip):

    Encoder                lib/encoder.py:356-432       ConvBlocksLegacy (lib/utils.py:928-947) + PatchMergingLegacy (:2173-2210) per stage
    bottleneck             MTL_model.py:196-206, :452-462   ConvBlocksLegacy -> TransformerEncoder (lib/vit_transformer.py:8694-8720, :8823-8878) -> ConvBlocksLegacy
    SegmentationDecoder    lib/decoder_alt.py:576-777   PatchExpandLegacy (lib/utils.py:1938-1963), SwinFilterBlock (lib/swin_cross_attention.py:114-178),
                                                         cat(skip, x), ConvBlocksLegacy
    inference              MTL_model.py:816-936          flip TTA on the softmax, optional Processor crop / un-crop

Same module and parameter names as the reference (its checkpoint's `state_dict` loads unchanged; running statistics included).
At inference BatchNorm is a per-channel affine map, folded into the convolution in front of it when the weights are loaded, so every
conv -> BN -> GELU is ONE launch of the f16-split MFMA kernel with the activation in its epilogue.  The windowed attention is
cf_window_attention; its projections, the skip gate and every 3x3 run on the kernels of the flow path.
`add_absolute_pos` / `init_weights` (required by MTLmodel.__init__, never passed by the reference's build_2d_model) are False / None.
"""
import torch

from . import ops
from .nn import Module, Conv2d, ConvTranspose2d, LayerNormCF, MultiheadAttention, _Linear, position_embedding_sine_2d


class BatchNorm2d(Module):
    """nn.BatchNorm2d in eval mode: parameters + running statistics; never run on its own (folded into the neighbouring convolution)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        for n in ("weight", "bias", "running_mean", "running_var"):
            self._param(n, (channels,))

    def scale_shift(self):
        s = self._p["weight"] / torch.sqrt(self._p["running_var"] + self.eps)
        return s, self._p["bias"] - self._p["running_mean"] * s


class Sequential(Module):
    """nn.Sequential slots by index ({index: Module}); state-dict names '<index>.<param>'."""

    def __init__(self, mods):
        super().__init__()
        object.__setattr__(self, "_mods", dict(mods))
        object.__setattr__(self, "_folded", {})

    def _children(self):
        for i, m in self._mods.items():
            yield str(i), m

    def conv_bn(self, i, x, x2=None, act="gelu", stride_obj=None):
        """conv at slot i followed by the BatchNorm at slot i + 1 and `act`, as one convolution with folded weights"""
        key = (i, ops.CONV_MODE)
        if i not in self._folded:
            conv, bn = self._mods[i], self._mods[i + 1]
            s, t = bn.scale_shift()
            f = Conv2d(conv.cin, conv.cout, conv.ks, stride=conv.stride, padding=conv.pad)
            f._p["weight"] = (conv._p["weight"] * s.view(-1, 1, 1, 1)).contiguous()
            f._p["bias"] = (conv._p["bias"] * s + t).contiguous()
            f._prepare()
            self._folded[i] = f
        del key
        return self._folded[i](x, x2=x2, act=act)

    def convT_bn_gelu(self, x):
        """ConvTranspose2d(k2, s2) at slot 0 + BatchNorm at slot 1 + GELU (PatchExpandLegacy.up)"""
        if "T" not in self._folded:
            ct, bn = self._mods[0], self._mods[1]
            s, t = bn.scale_shift()
            f = ConvTranspose2d(ct.cin, ct.cout)
            f._p["weight"] = (ct._p["weight"] * s.view(1, -1, 1, 1)).contiguous()
            f._p["bias"] = (ct._p["bias"] * s + t).contiguous()
            f._prepare()
            self._folded["T"] = f
        y = self._folded["T"](x)
        return ops.copy_channels(y, 0, y.shape[1], dst=y, act="gelu")          # in place: GELU of the folded ConvT + BN


class ConvBlocksLegacy(Module):
    """lib/utils.py:928-947: per block conv3x3 -> norm -> GELU -> conv3x3 -> norm -> GELU; widths torch.linspace(in, out, nb + 1).int()"""

    def __init__(self, in_dim, out_dim, nb_blocks):
        super().__init__()
        dims = torch.linspace(in_dim, out_dim, nb_blocks + 1).int().tolist()
        self.blocks = [Sequential({0: Conv2d(dims[i], dims[i + 1], 3, padding=1), 1: BatchNorm2d(dims[i + 1]),
                                   3: Conv2d(dims[i + 1], dims[i + 1], 3, padding=1), 4: BatchNorm2d(dims[i + 1])}) for i in range(nb_blocks)]

    def forward(self, x, x2=None):
        for j, b in enumerate(self.blocks):
            x = b.conv_bn(0, x, x2=x2 if j == 0 else None)
            x = b.conv_bn(3, x)
        return x


class PatchMergingLegacy(Module):
    """lib/utils.py:2173-2210: conv3x3 stride 2 -> norm -> GELU"""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.reduction = Sequential({0: Conv2d(in_dim, out_dim, 3, stride=2, padding=1), 1: BatchNorm2d(out_dim)})

    def forward(self, x):
        return self.reduction.conv_bn(0, x)


class PatchExpandLegacy(Module):
    """lib/utils.py:1938-1963: ConvTranspose2d(2, 2) -> norm -> GELU"""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.up = Sequential({0: ConvTranspose2d(in_dim, out_dim), 1: BatchNorm2d(out_dim)})

    def forward(self, x):
        return self.up.convT_bn_gelu(x)


class Encoder(Module):
    """lib/encoder.py:356-432"""

    def __init__(self, conv_depth, in_dims, out_dims):
        super().__init__()
        n = len(conv_depth)
        self.layers, self.downsample_layers = [], []
        for i in range(n):
            out_dim = 2 * out_dims[i] if i == n - 1 else in_dims[i + 1]
            self.layers.append(ConvBlocksLegacy(in_dims[i], out_dims[i], conv_depth[i]))
            self.downsample_layers.append(PatchMergingLegacy(out_dims[i], out_dim))

    def forward(self, x):
        skips = []
        for layer, ds in zip(self.layers, self.downsample_layers):
            x = layer(x)
            skips.append(x)
            x = ds(x)
        return x, skips


class _QKV(Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = _Linear(dim, 3 * dim)


class _BeforeCrossAttention(Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = LayerNormCF(dim)


class _CrossAttention(Module):
    """lib/swin_cross_attention.py:292-380: parameters of the window attention (same_key_query=True)"""

    def __init__(self, dim, window, heads):
        super().__init__()
        self._param("relative_position_bias_table", ((2 * window - 1) ** 2, heads))
        self.get_qkv_object_rescaled = _QKV(dim)
        self.get_qkv_object_rescaler = _QKV(dim)
        self.proj = _Linear(dim, dim)


class SwinCrossAttention(Module):
    """lib/swin_cross_attention.py:13-112: LayerNorm both inputs, q / k from the `rescaler` map and v from the `rescaled` one, windowed
    attention with relative position bias (shift_size 0 or window // 2), output projection; no residual, no MLP."""

    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size):
        super().__init__()
        self.dim, self.heads = dim, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.before_cross_attention_img1 = _BeforeCrossAttention(dim)
        self.before_cross_attention_img2 = _BeforeCrossAttention(dim)
        self.cross_attn = _CrossAttention(dim, self.window_size, num_heads)

    def _proj(self, lin, rows, x):
        """rows [a, b) of a Linear as a 1x1 convolution on [B,C,H,W]"""
        key = "_rows_%d_%d" % rows
        if key not in lin.__dict__:
            c = Conv2d(lin.cin, rows[1] - rows[0], 1)
            c._p["weight"] = lin._p["weight"][rows[0]:rows[1]].reshape(rows[1] - rows[0], lin.cin, 1, 1).contiguous()
            c._p["bias"] = lin._p["bias"][rows[0]:rows[1]].contiguous()
            c._prepare()
            lin.__dict__[key] = c
        return lin.__dict__[key](x)

    def forward(self, rescaled, rescaler):
        B, C, H, W = rescaled.shape
        g = self.before_cross_attention_img1.norm1(rescaled.view(B, C, H * W), inplace=False).view(B, C, H, W)
        x = self.before_cross_attention_img2.norm1(rescaler.view(B, C, H * W), inplace=False).view(B, C, H, W)
        qk = self._proj(self.cross_attn.get_qkv_object_rescaler.qkv, (0, 2 * C), x)
        v = self._proj(self.cross_attn.get_qkv_object_rescaled.qkv, (2 * C, 3 * C), g)
        a = ops.window_attention(qk, v, self.cross_attn._p["relative_position_bias_table"], self.heads, self.window_size, self.shift_size)
        return self._proj(self.cross_attn.proj, (0, C), a)


class SwinFilterBlock(Module):
    """lib/swin_cross_attention.py:114-178: filtered = skip * sigmoid(BN(conv1x1(attention blocks(W_g(skip), W_x(x)))))"""

    def __init__(self, in_dim, out_dim, input_resolution, num_heads, window_size, depth=2):
        super().__init__()
        self.W_g = Sequential({0: Conv2d(in_dim, out_dim, 1), 1: BatchNorm2d(out_dim)})
        self.W_x = Sequential({0: Conv2d(in_dim, out_dim, 1), 1: BatchNorm2d(out_dim)})
        self.blocks = [SwinCrossAttention(out_dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2) for i in range(depth)]
        self.psi = Sequential({0: Conv2d(out_dim, out_dim, 1), 1: BatchNorm2d(out_dim)})

    def forward(self, x, skip_co):
        g1, x1 = self.W_g.conv_bn(0, skip_co), self.W_x.conv_bn(0, x)
        for blk in self.blocks:
            g1 = blk(g1, x1)
        return ops.mul(skip_co, self.psi.conv_bn(0, g1, act="sigmoid"))


class _DeepSupervision(Module):
    def __init__(self, dim, num_classes):
        super().__init__()
        self.conv = Conv2d(dim, num_classes, 1)


class SegmentationDecoder(Module):
    """lib/decoder_alt.py:576-777 (deep supervision heads are loaded and unused: do_ds is False at inference)"""

    def __init__(self, conv_depth, spatial_cross_attention_num_heads, in_encoder_dims, out_encoder_dims, num_classes, window_size, img_size,
                 filter_skip_co_segmentation=True, deep_supervision=True):
        super().__init__()
        n = len(conv_depth)
        self.filter = filter_skip_co_segmentation
        self.layers, self.upsample_layers, self.encoder_skip_layers = [], [], []
        ds = {}
        for i in range(n):
            in_dim = out_encoder_dims[i] * 2 if i == 0 else in_encoder_dims[i - 1]
            res = img_size // (2 ** (n - i - 1))
            if filter_skip_co_segmentation:
                self.encoder_skip_layers.append(SwinFilterBlock(out_encoder_dims[i], out_encoder_dims[i], (res, res), spatial_cross_attention_num_heads[i],
                                                                window_size))
            self.upsample_layers.append(PatchExpandLegacy(in_dim, out_encoder_dims[i]))
            if deep_supervision and i != n - 1:
                ds[i] = _DeepSupervision(in_encoder_dims[i], num_classes)
            self.layers.append(ConvBlocksLegacy(out_encoder_dims[i] * 2, in_encoder_dims[i], conv_depth[i]))
        self.deep_supervision_layers = ds

    def forward(self, x, skips):
        for i, (layer_up, up, skip) in enumerate(zip(self.layers, self.upsample_layers, reversed(skips))):
            x = up(x)
            if self.filter:
                skip = self.encoder_skip_layers[i](x, skip)
            x = layer_up(skip, x2=x)                                # torch.cat((skip, x), 1) as a dual-input convolution
        return [x]


class TransformerEncoderLayer(Module):
    """lib/vit_transformer.py:8823-8878, post-norm: x = LN(x + MHA(q = k = x + pos, v = x)); x = LN(x + FFN_gelu(x)); channel-first tokens"""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.linear1, self.linear2 = _Linear(d_model, dim_feedforward), _Linear(dim_feedforward, d_model)
        self.norm1, self.norm2 = LayerNormCF(d_model), LayerNormCF(d_model)

    def forward(self, x, pos):
        B, C, N, _ = x.shape
        qp = ops.add(x, pos)
        x = self.self_attn(qp, qp, x, residual=x, same_qk=True)
        x = self.norm1(x.view(B, C, N)).view(B, C, N, 1)
        x = self.linear2(self.linear1(x, act="gelu"), res=x)
        return self.norm2(x.view(B, C, N)).view(B, C, N, 1)


class TransformerEncoder(Module):
    """lib/vit_transformer.py:8694-8720"""

    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = [TransformerEncoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)]

    def forward(self, x, pos):
        B, C, H, W = x.shape
        t = x.reshape(B, C, H * W, 1)
        for layer in self.layers:
            t = layer(t, pos)
        return t.view(B, C, H, W)


class MTLmodel(Module):
    """network_architecture/MTL_model.py:84-470; forward(x [B,1,H,W]) -> {'pred': logits [B,num_classes,H,W]} (the single-image branch
    :440-470 with do_ds False).  Constructor: the reference's keyword names for the values its YAML configs set."""

    def __init__(self, image_size, window_size, num_classes, in_dims=(1, 128, 256), out_encoder_dims=(64, 128, 256), conv_depth=(2, 2, 2),
                 spatial_cross_attention_num_heads=(2, 4, 8), bottleneck_heads=8, num_bottleneck_layers=1, asymmetric_unet=True,
                 filter_skip_co_segmentation=True, deep_supervision=True, processor=None):
        super().__init__()
        in_dims, out_encoder_dims, conv_depth = list(in_dims), list(out_encoder_dims), list(conv_depth)
        self.num_classes, self.image_size = num_classes, image_size
        self._processor = processor
        self.d_model = out_encoder_dims[-1] * 2
        self.encoder = Encoder(conv_depth, in_dims, out_encoder_dims)
        dec_depth = [x // 2 for x in conv_depth[::-1]] if asymmetric_unet else conv_depth[::-1]
        dec_out = in_dims[::-1]
        dec_out[-1] = num_classes
        self.decoder = SegmentationDecoder(dec_depth, list(spatial_cross_attention_num_heads)[::-1], dec_out, out_encoder_dims[::-1], num_classes,
                                           window_size, image_size, filter_skip_co_segmentation, deep_supervision)
        self.extra_bottleneck_block_1 = ConvBlocksLegacy(self.d_model, self.d_model, 1)
        self.bottleneck = TransformerEncoder(self.d_model, bottleneck_heads, 4 * self.d_model, num_bottleneck_layers)
        self.extra_bottleneck_block_2 = ConvBlocksLegacy(self.d_model, self.d_model, 1)

    def forward(self, x):
        x, skips = self.encoder(x)
        x = self.extra_bottleneck_block_1(x)
        B, C, H, W = x.shape
        x = self.bottleneck(x, position_embedding_sine_2d(H, W, C, x.device))
        x = self.extra_bottleneck_block_2(x)
        return {"pred": self.decoder(x, skips)[0]}

    # -- MTL_model.py:816-936 _internal_maybe_mirror_and_pred_2D
    def mirror_and_predict_2d(self, x, mirror_axes=(0, 1), do_mirroring=True, mult=None, normalize=False):
        """x [B,1,H,W] on the device -> flip-TTA softmax [B,K,H,W]; with a Processor attached every sample is cropped around its
        centroid first (preprocess_no_registration + crop_and_pad) and the result is zero-padded back (uncrop_no_registration);
        normalize=True z-scores every (cropped) sample first (NormalizeIntensity)."""
        from .inference import mirror_and_predict_2d
        proc = self._processor
        pads = None
        if proc is not None:
            crops, pads = [], []
            for b in range(x.shape[0]):
                cen, _ = proc.preprocess_no_registration(x[b][None].contiguous())
                c, p = proc.crop_and_pad(x[b][None].contiguous(), [int(v) for v in cen])
                crops.append(c)
                pads.append(p)
            x = torch.cat(crops, dim=0)
        if normalize:
            x = x.contiguous().clone()
            flat = x.view(x.shape[0], 1, -1)
            ops.group_norm(flat, None, None, 1, eps=0.0, out=flat)
        net = type("_", (), {"num_classes": self.num_classes, "__call__": lambda s, t: self(t)["pred"]})()
        out = mirror_and_predict_2d(net, x.contiguous(), mirror_axes, do_mirroring, None)
        if pads is not None:
            out = torch.stack([proc.uncrop_no_registration(out[b].contiguous(), pads[b]) for b in range(out.shape[0])], dim=0)
        if mult is not None:
            ops.mul(out, mult, out=out)
        return out
