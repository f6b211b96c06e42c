"""CPU restatement of the reference's `MTLmodel` segmenter / cropping network.  TEST INFRASTRUCTURE ONLY.

`nnunet/network_architecture/MTL_model.py:84-470` as the fork's own configs build it (`adversarial_acdc.yaml`, `seg_model.yaml`:
`transformer_depth: []`, `conv_depth: [2,2,2]`, `norm: BatchNorm2d`, `transformer_bottleneck`, `add_extra_bottleneck_blocks`,
`asymmetric_unet`, `filter_skip_co_segmentation`, `middle=False`):

    Encoder (lib/encoder.py:356-432): per stage ConvBlocksLegacy (lib/utils.py:928-947) + PatchMergingLegacy (:2173-2210)
    extra_bottleneck_block_1 -> TransformerEncoder (lib/vit_transformer.py:8694-8720, layer :8823-8878, post-norm, sine positions)
        -> extra_bottleneck_block_2
    SegmentationDecoder (lib/decoder_alt.py:576-777): per stage PatchExpandLegacy (lib/utils.py:1938-1963), SwinFilterBlock
        (lib/swin_cross_attention.py:114-178: two windowed cross-attention blocks, the second shifted by window // 2, q and k from the
        decoder feature, v from the skip; sigmoid gate on the skip), cat(skip, x), ConvBlocksLegacy

Module and parameter names follow the reference so that its `state_dict` loads with strict=True (tests/golden/make_golden_mtl.py pins
this file against the reference's own forward).  Two constructor arguments the reference's `build_2d_model` (lib/training_utils.py:1938-1996)
does not pass although `MTLmodel.__init__` requires them -- `add_absolute_pos`, `init_weights` -- are fixed at False / None.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def sine_pos_2d(B, H, W, num_pos_feats, temperature=10000, scale=2 * math.pi):
    """PositionEmbeddingSine2d(num_pos_feats, normalize=True) (lib/position_embedding.py:88-107) -> [B, 2*num_pos_feats, H, W]"""
    y_embed = torch.arange(1, H + 1, dtype=torch.float32)[None, :, None].repeat(B, 1, W)
    x_embed = torch.arange(1, W + 1, dtype=torch.float32)[None, None, :].repeat(B, H, 1)
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class ConvBlocksLegacy(nn.Module):
    """lib/utils.py:928-947"""

    def __init__(self, in_dim, out_dim, nb_blocks, norm=nn.BatchNorm2d, kernel_size=3):
        super().__init__()
        dims = torch.linspace(in_dim, out_dim, nb_blocks + 1).int().tolist()
        self.blocks = nn.ModuleList()
        for i in range(nb_blocks):
            self.blocks.append(nn.Sequential(nn.Conv2d(dims[i], dims[i + 1], kernel_size, padding="same"), norm(dims[i + 1]), nn.GELU(),
                                             nn.Conv2d(dims[i + 1], dims[i + 1], kernel_size, padding="same"), norm(dims[i + 1]), nn.GELU()))

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class PatchMergingLegacy(nn.Module):
    """lib/utils.py:2173-2210 (blur False, swin_abs_pos False)"""

    def __init__(self, in_dim, out_dim, norm=nn.BatchNorm2d):
        super().__init__()
        self.reduction = nn.Sequential(nn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1), norm(out_dim), nn.GELU())

    def forward(self, x):
        return self.reduction(x)


class PatchExpandLegacy(nn.Module):
    """lib/utils.py:1938-1963 (swin_abs_pos False)"""

    def __init__(self, in_dim, out_dim, norm=nn.BatchNorm2d):
        super().__init__()
        self.up = nn.Sequential(nn.ConvTranspose2d(in_dim, out_dim, 2, 2), norm(out_dim), nn.GELU())

    def forward(self, x):
        return self.up(x)


class Encoder(nn.Module):
    """lib/encoder.py:356-432"""

    def __init__(self, conv_depth, in_dims, out_dims, norm=nn.BatchNorm2d):
        super().__init__()
        n = len(conv_depth)
        self.layers, self.downsample_layers = nn.ModuleList(), nn.ModuleList()
        for i in range(n):
            out_dim = 2 * out_dims[i] if i == n - 1 else in_dims[i + 1]
            self.layers.append(ConvBlocksLegacy(in_dims[i], out_dims[i], conv_depth[i], norm))
            self.downsample_layers.append(PatchMergingLegacy(out_dims[i], out_dim, norm))

    def forward(self, x):
        skips = []
        for layer, ds in zip(self.layers, self.downsample_layers):
            x = layer(x)
            skips.append(x)
            x = ds(x)
        return x, skips


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class _QKV(nn.Module):
    """lib/swin_cross_attention.py get_qkv"""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)

    def forward(self, x):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        return qkv[0], qkv[1], qkv[2]


class CrossAttention(nn.Module):
    """lib/swin_cross_attention.py:292-380 (same_key_query=True: q and k from the `rescaler` windows, v from the `rescaled` ones)"""

    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.window_size, self.num_heads = window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.get_qkv_object_rescaled = _QKV(dim, num_heads)
        self.get_qkv_object_rescaler = _QKV(dim, num_heads)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(window_size[0]), torch.arange(window_size[1])], indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.proj = nn.Linear(dim, dim)

    def forward(self, rescaled, rescaler, mask=None):
        B_, N, C = rescaled.shape
        _, _, v = self.get_qkv_object_rescaled(rescaled)
        q, k, _ = self.get_qkv_object_rescaler(rescaler)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
        attn = attn + bias.unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = (attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, self.num_heads, N, N)
        attn = attn.softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


class BeforeCrossAttention(nn.Module):
    """lib/swin_cross_attention.py:589-620: LayerNorm, cyclic shift, window partition"""

    def __init__(self, dim, input_resolution, window_size, shift_size):
        super().__init__()
        self.window_size, self.input_resolution, self.shift_size = window_size, input_resolution, shift_size
        self.norm1 = nn.LayerNorm(dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        x = self.norm1(x).view(B, H, W, C)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        return window_partition(x, self.window_size).view(-1, self.window_size * self.window_size, C)


class SwinCrossAttention(nn.Module):
    """lib/swin_cross_attention.py:13-112"""

    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size):
        super().__init__()
        self.input_resolution, self.window_size, self.shift_size = input_resolution, window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.before_cross_attention_img1 = BeforeCrossAttention(dim, input_resolution, self.window_size, self.shift_size)
        self.before_cross_attention_img2 = BeforeCrossAttention(dim, input_resolution, self.window_size, self.shift_size)
        self.cross_attn = CrossAttention(dim, (self.window_size, self.window_size), num_heads)
        if self.shift_size > 0:
            H, W = input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            sl = (slice(0, -self.window_size), slice(-self.window_size, -self.shift_size), slice(-self.shift_size, None))
            cnt = 0
            for h in sl:
                for w in sl:
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = window_partition(img_mask, self.window_size).view(-1, self.window_size * self.window_size)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, rescaled, rescaler):
        B, C, H, W = rescaled.shape
        rescaled = rescaled.permute(0, 2, 3, 1).reshape(B, H * W, C)
        rescaler = rescaler.permute(0, 2, 3, 1).reshape(B, H * W, C)
        a = self.cross_attn(self.before_cross_attention_img1(rescaled), self.before_cross_attention_img2(rescaler), mask=self.attn_mask)
        x = window_reverse(a.view(-1, self.window_size, self.window_size, C), self.window_size, H, W)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        return x.view(B, H * W, C).permute(0, 2, 1).reshape(B, C, H, W)


class SwinFilterBlock(nn.Module):
    """lib/swin_cross_attention.py:114-178 (add_absolute_pos False)"""

    def __init__(self, in_dim, out_dim, input_resolution, num_heads, window_size, depth=2, norm=nn.BatchNorm2d):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv2d(in_dim, out_dim, 1), norm(out_dim), nn.GELU())
        self.W_x = nn.Sequential(nn.Conv2d(in_dim, out_dim, 1), norm(out_dim), nn.GELU())
        self.blocks = nn.ModuleList([SwinCrossAttention(out_dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2)
                                     for i in range(depth)])
        self.psi = nn.Sequential(nn.Conv2d(out_dim, out_dim, 1), norm(out_dim), nn.Sigmoid())

    def forward(self, x, skip_co):
        g1, x1 = self.W_g(skip_co), self.W_x(x)
        for blk in self.blocks:
            g1 = blk(g1, x1)
        return skip_co * self.psi(g1)


class DeepSupervision(nn.Module):
    """lib/utils.py:1813-1826 (only present in the state dict; unused at inference, do_ds False)"""

    def __init__(self, dim, num_classes):
        super().__init__()
        self.conv = nn.Conv2d(dim, num_classes, 1)

    def forward(self, x):
        return self.conv(x)


class SegmentationDecoder(nn.Module):
    """lib/decoder_alt.py:576-777"""

    def __init__(self, conv_depth, spatial_cross_attention_num_heads, in_encoder_dims, out_encoder_dims, num_classes, window_size, img_size,
                 filter_skip_co_segmentation=True, deep_supervision=True, norm=nn.BatchNorm2d):
        super().__init__()
        n = self.num_stages = len(conv_depth)
        self.filter = filter_skip_co_segmentation
        self.layers, self.upsample_layers = nn.ModuleList(), nn.ModuleList()
        self.deep_supervision_layers, self.encoder_skip_layers = nn.ModuleList(), nn.ModuleList()
        for i in range(n):
            in_dim = out_encoder_dims[i] * 2 if i == 0 else in_encoder_dims[i - 1]
            res = img_size // (2 ** (n - i - 1))
            self.encoder_skip_layers.append(SwinFilterBlock(out_encoder_dims[i], out_encoder_dims[i], (res, res), spatial_cross_attention_num_heads[i],
                                                            window_size, 2, norm) if filter_skip_co_segmentation else nn.Identity())
            self.upsample_layers.append(PatchExpandLegacy(in_dim, out_encoder_dims[i], norm))
            self.deep_supervision_layers.append(nn.Identity() if (i == n - 1 or not deep_supervision) else DeepSupervision(in_encoder_dims[i], num_classes))
            self.layers.append(ConvBlocksLegacy(out_encoder_dims[i] * 2, in_encoder_dims[i], conv_depth[i], norm))

    def forward(self, x, skips):
        for layer_up, up, skip, filt in zip(self.layers, self.upsample_layers, reversed(skips), self.encoder_skip_layers):
            x = up(x)
            if self.filter:
                skip = filt(x, skip)
            x = layer_up(torch.cat((skip, x), dim=1))
        return [x]


class TransformerEncoderLayer(nn.Module):
    """lib/vit_transformer.py:8823-8878 (post-norm, GELU)"""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.0, batch_first=True)
        self.linear1, self.linear2 = nn.Linear(d_model, dim_feedforward), nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)

    def forward(self, src, pos):
        q = k = src + pos
        src = self.norm1(src + self.self_attn(q, k, value=src)[0])
        return self.norm2(src + self.linear2(F.gelu(self.linear1(src))))


class TransformerEncoder(nn.Module):
    """lib/vit_transformer.py:8694-8720"""

    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)])

    def forward(self, src, pos):
        B, C, H, W = src.shape
        out = torch.flatten(src, start_dim=2).permute(0, 2, 1)
        for layer in self.layers:
            out = layer(out, pos)
        return out.permute(0, 2, 1).reshape(B, C, H, W)


class MTLmodel(nn.Module):
    """network_architecture/MTL_model.py:84-470, the single-image branch of forward (:440-470): {'pred': logits [B, num_classes, H, W]}"""

    def __init__(self, image_size, window_size, num_classes, in_dims=(1, 128, 256), out_encoder_dims=(64, 128, 256), conv_depth=(2, 2, 2),
                 spatial_cross_attention_num_heads=(2, 4, 8), bottleneck_heads=8, num_bottleneck_layers=1, asymmetric_unet=True,
                 filter_skip_co_segmentation=True, deep_supervision=True, norm=nn.BatchNorm2d):
        super().__init__()
        in_dims, out_encoder_dims, conv_depth = list(in_dims), list(out_encoder_dims), list(conv_depth)
        self.num_classes = num_classes
        self.d_model = out_encoder_dims[-1] * 2
        self.encoder = Encoder(conv_depth, in_dims, out_encoder_dims, norm)
        dec_depth = [x // 2 for x in conv_depth[::-1]] if asymmetric_unet else conv_depth[::-1]
        dec_out = in_dims[::-1]
        dec_out[-1] = num_classes
        self.decoder = SegmentationDecoder(dec_depth, list(spatial_cross_attention_num_heads)[::-1], dec_out, out_encoder_dims[::-1], num_classes,
                                           window_size, image_size, filter_skip_co_segmentation, deep_supervision, norm)
        self.extra_bottleneck_block_1 = ConvBlocksLegacy(self.d_model, self.d_model, 1, norm)
        self.bottleneck = TransformerEncoder(self.d_model, bottleneck_heads, 4 * self.d_model, num_bottleneck_layers)
        self.extra_bottleneck_block_2 = ConvBlocksLegacy(self.d_model, self.d_model, 1, norm)

    def forward(self, x):
        x, skips = self.encoder(x)
        x = self.extra_bottleneck_block_1(x)
        B, C, H, W = x.shape
        pos = torch.flatten(sine_pos_2d(B, H, W, self.d_model // 2), start_dim=2).permute(0, 2, 1)
        x = self.bottleneck(x, pos)
        x = self.extra_bottleneck_block_2(x)
        return {"pred": self.decoder(x, skips)[0]}
