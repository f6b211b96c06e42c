"""Oracle restatement of the reference's hot-path modules (CPU PyTorch, fp32).

Test infrastructure only -- see oracle/__init__.py.  Attribute names follow the
reference so that ``load_state_dict(reference.state_dict(), strict=True)``
works (tests/golden/make_golden.py checks exactly that).  `file:line`
citations are relative to /root/reference.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


# ----------------------------------------------------------------------------- integration.py
class SpatialTransformer(nn.Module):
    """nnunet/network_architecture/integration.py:37-79.  The `mode` call argument is ignored (:79)."""

    def __init__(self, size, mode="bilinear"):
        super().__init__()
        self.mode = mode
        self.register_buffer("grid", ops.identity_grid(size))

    def forward(self, flow, original, mode="bilinear"):
        return ops.warp_bilinear(flow, original, mode=self.mode)


class VecInt(nn.Module):
    """integration.py:82-99."""

    def __init__(self, inshape, nsteps):
        super().__init__()
        self.nsteps = nsteps
        self.scale = 1.0 / (2 ** nsteps)
        self.transformer = SpatialTransformer(inshape)

    def forward(self, vec):
        vec = vec * self.scale
        for _ in range(self.nsteps):
            vec = vec + self.transformer(vec, vec)
        return vec


# ----------------------------------------------------------------------------- convGRU.py
class ConvGRUCell(nn.Module):
    """nnunet/network_architecture/convGRU.py:7-69 (gamma = reset gate, beta = update gate)."""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size=(3, 3), bias=True):
        super().__init__()
        self.height, self.width = input_size
        pad = kernel_size[0] // 2, kernel_size[1] // 2
        self.hidden_dim = hidden_dim
        self.conv_gates = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, padding=pad, bias=bias)
        self.conv_can = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=pad, bias=bias)

    def forward(self, x, h):
        cc = self.conv_gates(torch.cat([x, h], dim=1))
        gamma, beta = torch.split(cc, self.hidden_dim, dim=1)
        r = torch.sigmoid(gamma)
        u = torch.sigmoid(beta)
        c = torch.tanh(self.conv_can(torch.cat([x, r * h], dim=1)))
        return (1 - u) * h + u * c


# ----------------------------------------------------------------------------- lib/utils.py blocks
class DoubleConv(nn.Module):
    """nnunet/lib/utils.py:1182-1215: residual added AFTER the second GELU."""

    def __init__(self, in_dim, out_dim, residual, stride=1, kernel_size=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_dim, out_dim, kernel_size, stride=stride, padding=1)
        self.norm1 = nn.GroupNorm(8, out_dim)
        self.conv2 = nn.Conv2d(out_dim, out_dim, kernel_size, padding=1)
        self.norm2 = nn.GroupNorm(8, out_dim)
        self.residual = residual
        if residual and (in_dim != out_dim or stride != 1):
            self.downsample = nn.Sequential(nn.Conv2d(in_dim, out_dim, kernel_size=1, stride=stride),
                                            nn.GroupNorm(8, out_dim))
        else:
            self.downsample = nn.Identity()

    def forward(self, x):
        r = x
        x = F.gelu(self.norm1(self.conv1(x)))
        x = F.gelu(self.norm2(self.conv2(x)))
        if self.residual:
            x = x + self.downsample(r)
        return x


class SingleConv(nn.Module):
    """nnunet/lib/utils.py:1239-1264: residual added BEFORE the GELU; downsample is a bare 1x1 conv."""

    def __init__(self, in_dim, out_dim, residual, stride=1, kernel_size=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_dim, out_dim, kernel_size, stride=stride, padding=1)
        self.norm1 = nn.GroupNorm(8, out_dim)
        self.residual = residual
        if residual and (in_dim != out_dim or stride != 1):
            self.downsample = nn.Conv2d(in_dim, out_dim, kernel_size=1, stride=stride)
        else:
            self.downsample = nn.Identity()

    def forward(self, x):
        r = x
        x = self.norm1(self.conv1(x))
        if self.residual:
            x = x + self.downsample(r)
        return F.gelu(x)


class ConvBlocks2DGroupLegacy(nn.Module):
    """nnunet/lib/utils.py:1345-1366 (channel counts rounded to multiples of 8 at :1349)."""

    def __init__(self, in_dim, out_dim, nb_blocks, stride=1, residual=False, kernel_size=3, nb_conv=2):
        super().__init__()
        dims = torch.linspace(in_dim, out_dim, nb_blocks + 1).int()
        dims[1:] = (torch.round(dims[1:] / 8) * 8).int()
        fn = DoubleConv if nb_conv == 2 else SingleConv
        self.blocks = nn.ModuleList(
            [fn(in_dim=int(dims[i]), out_dim=int(dims[i + 1]), residual=residual, stride=stride)
             for i in range(nb_blocks)])

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class PatchExpand2DGroup(nn.Module):
    """nnunet/lib/utils.py:1982-1994."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.up = nn.Sequential(nn.ConvTranspose2d(in_dim, out_dim, 2, 2), nn.GroupNorm(8, out_dim), nn.GELU())

    def forward(self, x):
        return self.up(x)


class PatchMerging2DGroup(nn.Module):
    """nnunet/lib/utils.py:2210-2229."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.reduction = nn.Sequential(nn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1),
                                       nn.GroupNorm(8, out_dim), nn.GELU())

    def forward(self, x):
        return self.reduction(x)


# ----------------------------------------------------------------------------- lib/encoder.py, decoder_alt.py
class Encoder2D(nn.Module):
    """nnunet/lib/encoder.py:541-660 (norm='group', legacy=True)."""

    def __init__(self, d_model, conv_depth, in_dims, out_dims, nb_conv, extra_block, residual, downsample_conv,
                 motion_appearance=False):
        super().__init__()
        self.num_stages = len(conv_depth)
        self.extra_block = extra_block
        self.motion_appearance = motion_appearance
        self.layers = nn.ModuleList()
        self.downsample_layers = nn.ModuleList()
        out_dim = None
        for i in range(self.num_stages):
            out_dim = d_model if i == self.num_stages - 1 else in_dims[i + 1]
            self.layers.append(ConvBlocks2DGroupLegacy(in_dim=in_dims[i], out_dim=out_dims[i],
                                                       nb_blocks=conv_depth[i], residual=residual, nb_conv=nb_conv))
            if downsample_conv == 2:
                ds = ConvBlocks2DGroupLegacy(in_dim=out_dims[i], out_dim=out_dim, nb_blocks=1, residual=residual,
                                             nb_conv=nb_conv, stride=2)
            else:
                ds = PatchMerging2DGroup(in_dim=out_dims[i], out_dim=out_dim)
            self.downsample_layers.append(ds)
        if extra_block or motion_appearance:
            self.out_conv = ConvBlocks2DGroupLegacy(in_dim=out_dim, out_dim=out_dim, nb_blocks=conv_depth[-1],
                                                    residual=residual, nb_conv=nb_conv)

    def forward(self, x):
        skips = []
        for layer, ds in zip(self.layers, self.downsample_layers):
            x = layer(x)
            skips.append(x)
            x = ds(x)
        if self.motion_appearance:  # EncoderMotionAppearance.forward, encoder.py:789-801
            return self.out_conv(x), x, skips
        if self.extra_block:
            x = self.out_conv(x)
        return x, skips


class Decoder2D(nn.Module):
    """nnunet/lib/decoder_alt.py:807-923 (norm='group', legacy=True, skip_co=True, deep_supervision=False)."""

    def __init__(self, d_model, conv_depth, in_encoder_dims, out_encoder_dims, num_classes, dot_multiplier,
                 nb_conv, residual):
        super().__init__()
        self.num_stages = len(conv_depth)
        self.layers = nn.ModuleList()
        self.upsample_layers = nn.ModuleList()
        self.deep_supervision_layers = nn.ModuleList()
        for i in range(self.num_stages):
            in_dim = d_model if i == 0 else in_encoder_dims[i - 1]
            self.upsample_layers.append(PatchExpand2DGroup(in_dim=in_dim, out_dim=out_encoder_dims[i]))
            self.layers.append(ConvBlocks2DGroupLegacy(in_dim=out_encoder_dims[i] * dot_multiplier,
                                                       out_dim=out_encoder_dims[i], nb_blocks=conv_depth[i],
                                                       nb_conv=nb_conv, residual=residual))
            self.deep_supervision_layers.append(nn.Identity())
        self.final_conv = nn.Conv2d(out_encoder_dims[-1], num_classes, kernel_size=3, padding=1)

    def forward(self, x, skips):
        for layer, up, skip in zip(self.layers, self.upsample_layers, reversed(skips)):
            x = up(x)
            x = torch.cat((skip, x), dim=1)
            x = layer(x)
        return self.final_conv(x)


# ----------------------------------------------------------------------------- transformers
def position_embedding_sine_2d(B, H, W, num_pos_feats, temperature=10000, scale=2 * math.pi):
    """nnunet/lib/position_embedding.py:88-107 (normalize=True) -> [B, 2*num_pos_feats, H, W]."""
    not_mask = torch.ones((B, H, W), dtype=torch.bool)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
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


class TransformerFlowLayer(nn.Module):
    """nnunet/lib/vit_transformer.py:1228-1270 (post-norm, GELU FFN, dropout 0)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, batch_first=True)
        self.cross_attn = nn.MultiheadAttention(d_model, nhead, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, query, key, value, query_pos, key_pos):
        q = k = query + query_pos
        query = self.norm1(query + self.self_attn(q, k, value=query)[0])
        t = self.cross_attn(query=query + query_pos, key=key + key_pos, value=value)[0]
        query = self.norm2(query + t)
        t = self.linear2(F.gelu(self.linear1(query)))
        return self.norm3(query + t)


class CrossAttentionLayer(nn.Module):
    """nnunet/lib/vit_transformer.py:5240-5287."""

    def __init__(self, dim, nhead, num_layers, dim_feedforward):
        super().__init__()
        self.dim = dim
        self.num_layers = num_layers
        self.bilateral_attention_layers = nn.ModuleList(
            [TransformerFlowLayer(dim, nhead, dim_feedforward) for _ in range(num_layers)])

    def forward(self, query, key, value):
        B, C, H, W = query.shape
        tok = lambda t: t.permute(0, 2, 3, 1).contiguous().view(B, H * W, C)
        pos = tok(position_embedding_sine_2d(B, H, W, C // 2))
        q, k, v = tok(query), tok(key), tok(value)
        for layer in self.bilateral_attention_layers:
            q = layer(q, k, v, pos, pos)
        return q.permute(0, 2, 1).contiguous().view(B, C, H, W)


class TransformerFlowEncoderSuccessiveNoEmb(nn.Module):
    """nnunet/lib/vit_transformer.py:3596-3641 (dim_feedforward default 2048)."""

    def __init__(self, dim, nhead, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.bilateral_attention_layers = nn.ModuleList(
            [TransformerFlowLayer(dim, nhead) for _ in range(num_layers)])

    def forward(self, unlabeled):
        T, B, C, H, W = unlabeled.shape
        u = unlabeled.permute(0, 1, 3, 4, 2).contiguous().view(T, B, H * W, C)
        pos = position_embedding_sine_2d(B, H, W, C // 2).permute(0, 2, 3, 1).contiguous().view(B, H * W, C)
        pos = pos[None].repeat(T - 1, 1, 1, 1).view((T - 1) * B, H * W, C)
        bwd = u[:-1].reshape((T - 1) * B, H * W, C)
        fwd = u[1:].reshape((T - 1) * B, H * W, C)
        for layer in self.bilateral_attention_layers:
            c0 = torch.cat([fwd, bwd], dim=0)
            c1 = torch.cat([bwd, fwd], dim=0)
            p = torch.cat([pos, pos], dim=0)
            c0 = layer(c0, c1, c1, p, p)
            fwd, bwd = torch.chunk(c0, 2, dim=0)
        return fwd.view(T - 1, B, H * W, C).permute(0, 1, 3, 2).contiguous().view(T - 1, B, C, H, W)


# ----------------------------------------------------------------------------- RAFT pieces (PARITY UNPINNED)
class CorrVolume(nn.Module):
    """nnunet/lib/raft.py is absent; spec in oracle/ops.py corr_volume."""

    def __init__(self, radius, stride):
        super().__init__()
        self.radius, self.stride = radius, stride

    def forward(self, cur, prev):
        return ops.corr_volume(cur, prev, self.radius, self.stride)


class CorrBlock:
    """nnunet/lib/raft_initial.py is absent; published RAFT core/corr.py CorrBlock."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.radius = radius
        self.pyramid = ops.corr_pyramid(ops.corr_allpairs(fmap1, fmap2), num_levels)

    def __call__(self, coords):
        return ops.corr_lookup(self.pyramid, coords, self.radius)


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        cor = F.relu(self.convc2(cor))
        flo = F.relu(self.convf1(flow))
        flo = F.relu(self.convf2(flo))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=256):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, x):
        for cz, cr, cq in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
            hx = torch.cat([h, x], dim=1)
            z = torch.sigmoid(cz(hx))
            r = torch.sigmoid(cr(hx))
            q = torch.tanh(cq(torch.cat([r * h, x], dim=1)))
            h = (1 - z) * h + z * q
        return h


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class BasicUpdateBlock(nn.Module):
    """Published RAFT core/update.py BasicUpdateBlock; call site SegFlowGaussian.py:942."""

    def __init__(self, hidden_dim=128, corr_levels=4, corr_radius=4):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, net, inp, corr, flow):
        mf = self.encoder(flow, corr)
        net = self.gru(net, torch.cat([inp, mf], dim=1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


# ----------------------------------------------------------------------------- SegFlowGaussian
class SegFlowGaussian(nn.Module):
    """nnunet/network_architecture/SegFlowGaussian.py:70-357 (__init__), built as
    nnunet/lib/training_utils.py:1460-1537 maps raft_config.yaml / video.yaml.

    dispatch (SegFlowGaussian.py:379-392):
      motion_appearance=True  -> forward_motion_appearance (:1813-1912)              [raft_config.yaml]
      motion_appearance=False -> forward_..._cost_volume_transformer_cat (:1330-1447) [video.yaml]
      raft=True (build-defined) -> forward_multi_task_flow_deformable_raft (:875-969)
    """

    def __init__(self, image_size, in_dims=(6, 128, 256), out_encoder_dims=(64, 128, 256), d_model=256,
                 conv_depth=(1, 1, 1), skip_co_depth=(1, 1, 1), bottleneck_heads=4, nb_layers=1,
                 dim_feedforward=3072, motion_appearance=True, radius=(4, 4, 4, 4), stride=(4, 2, 1, 1),
                 nb_conv=2, residual=True, extra_block=True, downsample_conv=2, raft=False, raft_iters=12):
        super().__init__()
        in_dims = list(in_dims)
        out_encoder_dims = list(out_encoder_dims)
        conv_depth = list(conv_depth)
        self.num_stages = len(conv_depth)
        self.d_model = d_model
        self.image_size = image_size
        self.motion_appearance = motion_appearance
        self.raft = raft
        self.raft_iters = raft_iters
        self.H = self.W = int(image_size / 2 ** self.num_stages)

        self.integration = VecInt((image_size, image_size), 7)
        self.motion_estimation = SpatialTransformer((image_size, image_size))

        in_past = copy.copy(in_dims)
        in_past[0] = 6
        enc = dict(d_model=d_model, out_dims=out_encoder_dims, conv_depth=conv_depth, nb_conv=nb_conv,
                   residual=residual, downsample_conv=downsample_conv)
        self.memory_encoder = Encoder2D(in_dims=in_past, extra_block=extra_block, **enc)
        in_q = copy.copy(in_dims)
        self.skip_co_reduction_list = nn.ModuleList()
        if not motion_appearance:
            in_q[0] = 1
            self.query_encoder = Encoder2D(in_dims=in_q, extra_block=extra_block, **enc)
            self.cost_volume_encoder_list = nn.ModuleList()
            self.cost_volume_computation_list = nn.ModuleList()
            for idx, (dim, nb) in enumerate(zip(out_encoder_dims, skip_co_depth)):
                corr_dim = (2 * radius[idx] + 1) ** 2
                self.cost_volume_computation_list.append(CorrVolume(radius=radius[idx], stride=stride[idx]))
                self.cost_volume_encoder_list.append(
                    ConvBlocks2DGroupLegacy(in_dim=corr_dim, out_dim=dim, nb_blocks=1, residual=residual))
                self.skip_co_reduction_list.append(
                    ConvBlocks2DGroupLegacy(in_dim=2 * dim, out_dim=dim, nb_blocks=nb, residual=residual))
        else:
            in_q[0] = 2
            self.query_encoder = Encoder2D(in_dims=in_q, extra_block=False, motion_appearance=True, **enc)
            for dim, nb in zip(out_encoder_dims, skip_co_depth):
                self.skip_co_reduction_list.append(
                    ConvBlocks2DGroupLegacy(in_dim=2 * dim, out_dim=dim, nb_blocks=nb, residual=residual))

        dec_in = in_dims[:]
        dec_in[0] = 4
        self.flow_decoder = Decoder2D(d_model=d_model, dot_multiplier=2, conv_depth=conv_depth[::-1],
                                      in_encoder_dims=dec_in[::-1], out_encoder_dims=out_encoder_dims[::-1],
                                      num_classes=2, nb_conv=nb_conv, residual=residual)
        self.gru_cell = ConvGRUCell(input_size=(self.H, self.W), input_dim=d_model, hidden_dim=d_model)
        self.reduce_transformer = ConvBlocks2DGroupLegacy(in_dim=d_model * 2, out_dim=d_model, nb_blocks=1,
                                                          residual=residual)
        self.bottleneck1 = CrossAttentionLayer(dim=d_model, nhead=bottleneck_heads, num_layers=nb_layers,
                                               dim_feedforward=dim_feedforward)
        self.bottleneck2 = CrossAttentionLayer(dim=d_model, nhead=bottleneck_heads, num_layers=nb_layers,
                                               dim_feedforward=dim_feedforward)
        if raft:
            # never constructed by the reference; build-defined, published RAFT sizes
            self.update_block = BasicUpdateBlock(hidden_dim=d_model // 2)

    # -- shared tail of one recurrence step (SegFlowGaussian.py:1400-1435 == :1862-1905)
    def _memory_input(self, x0, xt, cum):
        reg = self.motion_estimation(flow=cum, original=xt)
        err = x0 - reg
        return torch.cat([x0, xt, cum, err, reg], dim=1)

    def forward(self, x):
        if self.raft:
            return self.forward_raft(x)
        if self.motion_appearance:
            return self.forward_motion_appearance(x)
        return self.forward_cost_volume(x)

    def forward_motion_appearance(self, x):
        """SegFlowGaussian.py:1813-1912."""
        T, B, C, H, W = x.shape
        cum = torch.zeros(B, 2, H, W)
        hidden = torch.zeros(B, self.d_model, self.H, self.W)
        past_motion, past_skips = self.memory_encoder(self._memory_input(x[0], x[0], cum))
        first_app, _, _ = self.query_encoder(torch.cat([x[0], x[0]], dim=1))
        prev_app = first_app
        flows = []
        for t in range(1, T):
            cur_app, _cur_motion, skips = self.query_encoder(torch.cat([x[t], x[t - 1]], dim=1))
            new_skips = [self.skip_co_reduction_list[s](torch.cat([skips[s], past_skips[s]], dim=1))
                         for s in range(self.num_stages)]
            f1 = self.bottleneck1(query=cur_app, key=prev_app, value=prev_app)
            f2 = self.bottleneck2(query=cur_app, key=first_app, value=past_motion)
            gru_in = self.reduce_transformer(torch.cat([f1, f2], dim=1))
            hidden = self.gru_cell(gru_in, hidden)
            flow = self.flow_decoder(hidden, new_skips)
            cum = cum + flow
            flows.append(cum)
            past_motion, past_skips = self.memory_encoder(self._memory_input(x[0], x[t], cum))
            prev_app = cur_app
        return {"backward_flow": torch.stack(flows, dim=0)}

    def forward_cost_volume(self, x):
        """SegFlowGaussian.py:1330-1447 (skip_co_type='both', correlation_value=False, warp=False)."""
        T, B, C, H, W = x.shape
        cum = torch.zeros(B, 2, H, W)
        hidden = torch.zeros(B, self.d_model, self.H, self.W)
        past_motion, past_skips = self.memory_encoder(self._memory_input(x[0], x[0], cum))
        first_feat, first_skips = self.query_encoder(x[0])
        prev_feat, prev_skips = first_feat, first_skips
        flows = []
        for t in range(1, T):
            cur_feat, cur_skips = self.query_encoder(x[t])
            new_skips = []
            for s in range(self.num_stages):
                corr = self.cost_volume_computation_list[s](cur_skips[s], prev_skips[s])
                corr = self.cost_volume_encoder_list[s](corr)
                new_skips.append(self.skip_co_reduction_list[s](torch.cat([corr, past_skips[s]], dim=1)))
            f1 = self.bottleneck1(query=cur_feat, key=prev_feat, value=prev_feat)
            f2 = self.bottleneck2(query=cur_feat, key=first_feat, value=past_motion)
            gru_in = self.reduce_transformer(torch.cat([f1, f2], dim=1))
            hidden = self.gru_cell(gru_in, hidden)
            flow = self.flow_decoder(hidden, new_skips)
            cum = cum + flow
            flows.append(cum)
            past_motion, past_skips = self.memory_encoder(self._memory_input(x[0], x[t], cum))
            prev_feat, prev_skips = cur_feat, cur_skips
        return {"backward_flow": torch.stack(flows, dim=0)}

    def forward_raft(self, x):
        """SegFlowGaussian.py:875-969.  The reference calls `self.update_block` (never built) and splits the
        tuple the encoders return; the build takes element [0] (the H/8 feature map) -- DESIGN.md."""
        T, B, C, H, W = x.shape
        flow_up = torch.zeros(B, 2, H, W)
        cnet = self.memory_encoder(self._memory_input(x[0], x[0], flow_up))[0]
        net, inp = torch.split(cnet, [self.d_model // 2, self.d_model // 2], dim=1)
        net = torch.tanh(net)
        inp = torch.relu(inp)
        coords0 = ops.coords_grid(B, H // 8, W // 8)
        coords1 = ops.coords_grid(B, H // 8, W // 8)
        f1 = self.query_encoder(x[0])[0]
        out = []
        for t in range(1, T):
            f2 = self.query_encoder(x[t])[0]
            corr_fn = CorrBlock(f1, f2, radius=4)
            it = []
            for _ in range(self.raft_iters):
                corr = corr_fn(coords1)
                flow = coords1 - coords0
                net, up_mask, delta = self.update_block(net, inp, corr, flow)
                coords1 = coords1 + delta
                flow_up = ops.convex_upsample(coords1 - coords0, up_mask)
                it.append(flow_up)
            out.append(torch.stack(it, dim=0))
            cnet = self.memory_encoder(self._memory_input(x[0], x[t], flow_up))[0]
            inp = torch.relu(torch.split(cnet, [self.d_model // 2, self.d_model // 2], dim=1)[1])
        return {"backward_flow": torch.stack(out, dim=1)}


# ----------------------------------------------------------------------------- successive model
class OpticalFlowModelSuccessive(nn.Module):
    """nnunet/network_architecture/Optical_flow_model_successive.py:186-404 (successive.yaml:
    downsample_conv=1, residual=False, no extra block, d_model = 2*out_dims[-1], 8 heads, FFN 2048)."""

    def __init__(self, image_size, nb_channels, in_dims=(6, 128, 256), out_encoder_dims=(64, 128, 256),
                 conv_depth=(1, 1, 1), bottleneck_heads=8, nb_layers=1, nb_conv=2, downsample_conv=1):
        super().__init__()
        in_dims = list(in_dims)
        out_encoder_dims = list(out_encoder_dims)
        conv_depth = list(conv_depth)
        self.num_stages = len(conv_depth)
        self.d_model = out_encoder_dims[-1] * 2
        self.image_size = image_size
        self.integration = VecInt((image_size, image_size), 7)
        in_dims[0] = nb_channels
        self.encoder = Encoder2D(d_model=self.d_model, out_dims=out_encoder_dims, in_dims=in_dims,
                                 conv_depth=conv_depth, nb_conv=nb_conv, extra_block=False, residual=False,
                                 downsample_conv=downsample_conv)
        dec_in = in_dims[:]
        dec_in[0] = 4
        self.flow_decoder = Decoder2D(d_model=self.d_model, dot_multiplier=2, conv_depth=conv_depth[::-1],
                                      in_encoder_dims=dec_in[::-1], out_encoder_dims=out_encoder_dims[::-1],
                                      num_classes=2, nb_conv=nb_conv, residual=False)
        self.bottleneck = TransformerFlowEncoderSuccessiveNoEmb(dim=self.d_model, nhead=bottleneck_heads,
                                                                num_layers=nb_layers)
        self.skip_co_reduction_list = nn.ModuleList(
            [ConvBlocks2DGroupLegacy(in_dim=2 * d, out_dim=d, nb_blocks=1, nb_conv=nb_conv)
             for d in out_encoder_dims])

    def forward(self, unlabeled, inference=False):
        feats, skips = [], []
        for t in range(len(unlabeled)):
            f, s = self.encoder(unlabeled[t])
            feats.append(f)
            skips.append(s)
        fwd = self.bottleneck(torch.stack(feats, dim=0))
        flows = []
        for t in range(len(fwd)):
            sk = [self.skip_co_reduction_list[s](torch.cat([skips[t][s], skips[t + 1][s]], dim=1))
                  for s in range(self.num_stages)]
            flows.append(self.flow_decoder(fwd[t], sk))
        flow = torch.stack(flows, dim=0)
        if inference:
            flow = torch.stack([self.integration(flow[t]) for t in range(len(flow))], dim=0)
        return {"flow": flow}


class ModelWrap(nn.Module):
    """Optical_flow_model_successive.py:58-134 (forward_from_ed, no_error=False)."""

    def __init__(self, model1, model2):
        super().__init__()
        self.model1 = model1
        self.model2 = model2
        self.motion_estimation = SpatialTransformer((model1.image_size, model1.image_size))

    def forward(self, x, inference=False):
        out2 = {}
        out1 = self.model1(x)
        if len(x) == 2:
            out2["flow"] = out1["flow"][0]
            return out1, out2
        flow1 = out1["flow"]
        cum = flow1[0]
        cums = [cum]
        for t in range(1, len(flow1)):
            reg1 = self.motion_estimation(flow=cum, original=x[t])
            reg2 = self.motion_estimation(flow=flow1[t], original=x[t + 1])
            err1 = x[0] - reg1
            err2 = x[t] - reg2
            x1 = torch.cat([cum, x[t], x[0], reg1, err1], dim=1)
            x2 = torch.cat([flow1[t], x[t + 1], x[t], reg2, err2], dim=1)
            out = self.model2(torch.stack([x1, x2], dim=0), inference=inference)
            cum = cum + out["flow"][0]
            cums.append(cum)
        out2["flow"] = cum
        out2["cumulated"] = torch.stack(cums, dim=0)
        return out1, out2


# ----------------------------------------------------------------------------- Generic_UNet (2D)
class ConvDropoutNormNonlin(nn.Module):
    """nnunet/network_architecture/generic_UNet.py:26-69 (dropout p=0 -> absent)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True)
        self.instnorm = nn.InstanceNorm2d(cout, eps=1e-5, affine=True)

    def forward(self, x):
        return F.leaky_relu(self.instnorm(self.conv(x)), 0.01)


class StackedConvLayers(nn.Module):
    """generic_UNet.py:79-144."""

    def __init__(self, cin, cout, num_convs, first_stride=None):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = nn.Sequential(
            *([ConvDropoutNormNonlin(cin, cout, first_stride if first_stride is not None else 1)] +
              [ConvDropoutNormNonlin(cout, cout) for _ in range(num_convs - 1)]))

    def forward(self, x):
        return self.blocks(x)


class GenericUNet2D(nn.Module):
    """generic_UNet.py:167-408 as instantiated at nnunet/training/network_training/nnUNetTrainerV2.py:147-169:
    InstanceNorm(affine), LeakyReLU(0.01), convolutional pooling, transposed-conv upsampling (no bias),
    1x1 seg heads without bias, final_nonlin = identity.  Returns the full-resolution logits."""

    MAX_FILTERS_2D = 480

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2, pool_op_kernel_sizes=None):
        """pool_op_kernel_sizes: per-stage pooling kernels from the plans (generic_UNet.py:247-248, first_stride :283-285, transposed
        convolutions :343-344), e.g. [[2,2]]*5 + [[2,1]] for the ACDC 2-D patch (256, 224); default [[2,2]] * num_pool."""
        super().__init__()
        self.num_classes = num_classes
        pool = [tuple(int(v) for v in p_) for p_ in (pool_op_kernel_sizes or [(2, 2)] * num_pool)]
        assert len(pool) == num_pool
        ctx, loc, tu, seg = [], [], [], []
        out_f, in_f = base_num_features, input_channels
        for d in range(num_pool):
            ctx.append(StackedConvLayers(in_f, out_f, num_conv_per_stage, pool[d - 1] if d != 0 else None))
            in_f = out_f
            out_f = min(int(np.round(out_f * 2)), self.MAX_FILTERS_2D)
        final = out_f
        ctx.append(nn.Sequential(StackedConvLayers(in_f, out_f, num_conv_per_stage - 1, pool[-1]),
                                 StackedConvLayers(out_f, final, 1)))
        for u in range(num_pool):
            from_down = final
            from_skip = ctx[-(2 + u)].output_channels
            final = from_skip
            tu.append(nn.ConvTranspose2d(from_down, from_skip, pool[-(u + 1)], pool[-(u + 1)], bias=False))
            loc.append(nn.Sequential(StackedConvLayers(from_skip * 2, from_skip, num_conv_per_stage - 1),
                                     StackedConvLayers(from_skip, final, 1)))
        for ds in range(len(loc)):
            seg.append(nn.Conv2d(loc[ds][-1].output_channels, num_classes, 1, 1, 0, 1, 1, False))
        self.conv_blocks_localization = nn.ModuleList(loc)
        self.conv_blocks_context = nn.ModuleList(ctx)
        self.td = nn.ModuleList([])
        self.tu = nn.ModuleList(tu)
        self.seg_outputs = nn.ModuleList(seg)

    def forward(self, x):
        skips = []
        for d in range(len(self.conv_blocks_context) - 1):
            x = self.conv_blocks_context[d](x)
            skips.append(x)
        x = self.conv_blocks_context[-1](x)
        for u in range(len(self.tu)):
            x = self.tu[u](x)
            x = torch.cat((x, skips[-(u + 1)]), dim=1)
            x = self.conv_blocks_localization[u](x)
        return self.seg_outputs[-1](x)



# ----------------------------------------------------------------------------- Generic_UNet (3D)
class ConvDropoutNormNonlin3D(nn.Module):
    """generic_UNet.py:26-69 with conv_op = nn.Conv3d, InstanceNorm3d(affine), LeakyReLU(0.01); kernel 3 -> pad 1, 1 -> 0."""

    def __init__(self, cin, cout, kernel=(3, 3, 3), stride=(1, 1, 1)):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, tuple(kernel), stride=tuple(stride), padding=tuple(1 if k == 3 else 0 for k in kernel), bias=True)
        self.instnorm = nn.InstanceNorm3d(cout, eps=1e-5, affine=True)

    def forward(self, x):
        return F.leaky_relu(self.instnorm(self.conv(x)), 0.01)


class StackedConvLayers3D(nn.Module):
    """generic_UNet.py:79-144."""

    def __init__(self, cin, cout, num_convs, kernel, first_stride=None):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = nn.Sequential(
            *([ConvDropoutNormNonlin3D(cin, cout, kernel, first_stride if first_stride is not None else (1, 1, 1))] +
              [ConvDropoutNormNonlin3D(cout, cout, kernel) for _ in range(num_convs - 1)]))

    def forward(self, x):
        return self.blocks(x)


class GenericUNet3D(nn.Module):
    """generic_UNet.py:167-408 with conv_op = nn.Conv3d (the network behind _internal_predict_3D_3Dconv_tiled):
    per-stage pool_op_kernel_sizes / conv_kernel_sizes as in the plans (anisotropic (1,2,2) / (1,3,3) stages allowed),
    convolutional pooling and upsampling, MAX_NUM_FILTERS_3D = 320.  Returns the full-resolution logits."""

    MAX_NUM_FILTERS_3D = 320

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2,
                 pool_op_kernel_sizes=None, conv_kernel_sizes=None):
        super().__init__()
        self.num_classes = num_classes
        pool = [tuple(p) for p in (pool_op_kernel_sizes or [(2, 2, 2)] * num_pool)]
        kern = [tuple(k) for k in (conv_kernel_sizes or [(3, 3, 3)] * (num_pool + 1))]
        self.pool_op_kernel_sizes, self.conv_kernel_sizes = pool, kern
        ctx, loc, tu, seg = [], [], [], []
        out_f, in_f = base_num_features, input_channels
        for d in range(num_pool):
            ctx.append(StackedConvLayers3D(in_f, out_f, num_conv_per_stage, kern[d], pool[d - 1] if d != 0 else None))
            in_f = out_f
            out_f = min(int(np.round(out_f * 2)), self.MAX_NUM_FILTERS_3D)
        final = out_f
        ctx.append(nn.Sequential(StackedConvLayers3D(in_f, out_f, num_conv_per_stage - 1, kern[num_pool], pool[-1]),
                                 StackedConvLayers3D(out_f, final, 1, kern[num_pool])))
        for u in range(num_pool):
            from_down = final
            from_skip = ctx[-(2 + u)].output_channels
            final = from_skip
            tu.append(nn.ConvTranspose3d(from_down, from_skip, pool[-(u + 1)], pool[-(u + 1)], bias=False))
            loc.append(nn.Sequential(StackedConvLayers3D(from_skip * 2, from_skip, num_conv_per_stage - 1, kern[-(u + 1)]),
                                     StackedConvLayers3D(from_skip, final, 1, kern[-(u + 1)])))
        for ds in range(len(loc)):
            seg.append(nn.Conv3d(loc[ds][-1].output_channels, num_classes, 1, 1, 0, 1, 1, False))
        self.conv_blocks_localization = nn.ModuleList(loc)
        self.conv_blocks_context = nn.ModuleList(ctx)
        self.td = nn.ModuleList([])
        self.tu = nn.ModuleList(tu)
        self.seg_outputs = nn.ModuleList(seg)

    def forward(self, x):
        skips = []
        for d in range(len(self.conv_blocks_context) - 1):
            x = self.conv_blocks_context[d](x)
            skips.append(x)
        x = self.conv_blocks_context[-1](x)
        for u in range(len(self.tu)):
            x = self.tu[u](x)
            x = torch.cat((x, skips[-(u + 1)]), dim=1)
            x = self.conv_blocks_localization[u](x)
        return self.seg_outputs[-1](x)


# ----------------------------------------------------------------------------- sliding-window inference
def mirror_and_predict_2d(net, x, mirror_axes=(0, 1), do_mirroring=True, mult=None):
    """SegmentationNetwork._internal_maybe_mirror_and_pred_2D, neural_network.py:573-621.
    net: callable [B,C,X,Y] -> logits [B,K,X,Y]; softmax over dim 1 is the inference nonlinearity."""
    result = torch.zeros([x.shape[0], net.num_classes] + list(x.shape[2:]), dtype=torch.float)
    n = 2 ** len(mirror_axes) if do_mirroring else 1
    for m in range(4 if do_mirroring else 1):
        if m == 0:
            result += 1 / n * torch.softmax(net(x), 1)
        if m == 1 and (1 in mirror_axes):
            result += 1 / n * torch.flip(torch.softmax(net(torch.flip(x, (3,))), 1), (3,))
        if m == 2 and (0 in mirror_axes):
            result += 1 / n * torch.flip(torch.softmax(net(torch.flip(x, (2,))), 1), (2,))
        if m == 3 and (0 in mirror_axes) and (1 in mirror_axes):
            result += 1 / n * torch.flip(torch.softmax(net(torch.flip(x, (3, 2))), 1), (3, 2))
    if mult is not None:
        result[:, :] *= mult
    return result


def predict_2d_tiled(net, x, patch_size, step_size=0.5, do_mirroring=True, mirror_axes=(0, 1),
                     use_gaussian=True, pad_border_mode="constant", pad_kwargs=None):
    """SegmentationNetwork._internal_predict_2D_2Dconv_tiled, neural_network.py:623-769 (upstream
    nnU-Net 4-argument semantics).  x: numpy [C,X,Y] -> (seg [X,Y], softmax [K,X,Y])."""
    data, slicer = ops.pad_nd_image(x, patch_size, pad_border_mode, pad_kwargs, True, None)
    steps = ops.compute_steps_for_sliding_window(patch_size, data.shape[1:], step_size)
    num_tiles = len(steps[0]) * len(steps[1])
    if use_gaussian and num_tiles > 1:
        g = ops.get_gaussian(patch_size, sigma_scale=1.0 / 8)
        gauss_t = torch.from_numpy(g)
        add = g
    else:
        gauss_t = None
        add = np.ones(patch_size, dtype=np.float32)
    agg = np.zeros([net.num_classes] + list(data.shape[1:]), dtype=np.float32)
    cnt = np.zeros([net.num_classes] + list(data.shape[1:]), dtype=np.float32)
    for lx in steps[0]:
        for ly in steps[1]:
            tile = torch.from_numpy(np.ascontiguousarray(data[None, :, lx:lx + patch_size[0], ly:ly + patch_size[1]]))
            pred = mirror_and_predict_2d(net, tile, mirror_axes, do_mirroring, gauss_t)[0].numpy()
            agg[:, lx:lx + patch_size[0], ly:ly + patch_size[1]] += pred
            cnt[:, lx:lx + patch_size[0], ly:ly + patch_size[1]] += add
    sl = tuple([slice(0, agg.shape[i]) for i in range(len(agg.shape) - (len(slicer) - 1))] + slicer[1:])
    probs = agg[sl] / cnt[sl]
    return probs.argmax(0), probs


def predict_3d_2dconv_tiled(net, x, patch_size, **kw):
    """SegmentationNetwork._internal_predict_3D_2Dconv_tiled, neural_network.py:814-857: loop over z slices.
    x: numpy [C,Z,X,Y] -> (seg [Z,X,Y], softmax [K,Z,X,Y])."""
    segs, probs = [], []
    for z in range(x.shape[1]):
        s, p = predict_2d_tiled(net, x[:, z], patch_size, **kw)
        segs.append(s[None])
        probs.append(p[None])
    return np.vstack(segs), np.vstack(probs).transpose((1, 0, 2, 3))



def mirror_and_predict_3d(net, x, mirror_axes=(0, 1, 2), do_mirroring=True, mult=None):
    """SegmentationNetwork._internal_maybe_mirror_and_pred_3D, neural_network.py:506-571.
    net: callable [B,C,X,Y,Z] -> logits [B,K,X,Y,Z]; the result tensor is [1,K,...] and broadcasts like the reference's."""
    result = torch.zeros([1, net.num_classes] + list(x.shape[2:]), dtype=torch.float)
    n = 2 ** len(mirror_axes) if do_mirroring else 1
    sm = lambda t: torch.softmax(t, 1)
    for m in range(8 if do_mirroring else 1):
        if m == 0:
            result = result + 1 / n * sm(net(x))
        if m == 1 and (2 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (4,)))), (4,))
        if m == 2 and (1 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (3,)))), (3,))
        if m == 3 and (2 in mirror_axes) and (1 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (4, 3)))), (4, 3))
        if m == 4 and (0 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (2,)))), (2,))
        if m == 5 and (0 in mirror_axes) and (2 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (4, 2)))), (4, 2))
        if m == 6 and (0 in mirror_axes) and (1 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (3, 2)))), (3, 2))
        if m == 7 and (0 in mirror_axes) and (1 in mirror_axes) and (2 in mirror_axes):
            result = result + 1 / n * torch.flip(sm(net(torch.flip(x, (4, 3, 2)))), (4, 3, 2))
    if mult is not None:
        result[:, :] *= mult
    return result


def predict_3d_tiled(net, x, patch_size, step_size=0.5, do_mirroring=True, mirror_axes=(0, 1, 2), use_gaussian=True,
                     pad_border_mode="constant", pad_kwargs=None):
    """SegmentationNetwork._internal_predict_3D_3Dconv_tiled, neural_network.py:292-430 (all_in_gpu=False branch).
    x: numpy [C,X,Y,Z] -> (seg [X,Y,Z], softmax [K,X,Y,Z])."""
    assert len(x.shape) == 4, "x must be (c, x, y, z)"
    data, slicer = ops.pad_nd_image(x, patch_size, pad_border_mode, pad_kwargs, True, None)
    steps = ops.compute_steps_for_sliding_window(patch_size, data.shape[1:], step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    if use_gaussian and num_tiles > 1:
        g = ops.get_gaussian(patch_size, sigma_scale=1.0 / 8)
        gauss_t = torch.from_numpy(g)
        add = g
    else:
        gauss_t = None
        add = np.ones(patch_size, dtype=np.float32)
    agg = np.zeros([net.num_classes] + list(data.shape[1:]), dtype=np.float32)
    cnt = np.zeros([net.num_classes] + list(data.shape[1:]), dtype=np.float32)
    px, py, pz = patch_size
    for lx in steps[0]:
        for ly in steps[1]:
            for lz in steps[2]:
                tile = torch.from_numpy(np.ascontiguousarray(data[None, :, lx:lx + px, ly:ly + py, lz:lz + pz]))
                pred = mirror_and_predict_3d(net, tile, mirror_axes, do_mirroring, gauss_t)[0].numpy()
                agg[:, lx:lx + px, ly:ly + py, lz:lz + pz] += pred
                cnt[:, lx:lx + px, ly:ly + py, lz:lz + pz] += add
    sl = tuple([slice(0, agg.shape[i]) for i in range(len(agg.shape) - (len(slicer) - 1))] + slicer[1:])
    probs = agg[sl] / cnt[sl]
    return probs.argmax(0), probs


# ----------------------------------------------------------------------------- Processor crop arithmetic
def masks_to_boxes(masks):
    """torchvision.ops.masks_to_boxes (torchvision absent -> restated from its documentation, parity unpinned): masks [N,H,W] ->
    float [N,4] = (x1, y1, x2, y2), the extreme coordinates of the non-zero pixels."""
    out = torch.zeros((masks.shape[0], 4), dtype=torch.float32)
    for i, m in enumerate(masks):
        y, x = torch.where(m != 0)
        out[i] = torch.tensor([x.min(), y.min(), x.max(), y.max()], dtype=torch.float32)
    return out


class Processor:
    """nnunet/training/network_training/processor.py:109-138,178-186,223-230 (crop / uncrop arithmetic) and :140-176, :232-237
    (discretize / get_mean_centroid / preprocess_no_registration)."""

    def __init__(self, crop_size, image_size, cropping_network=None):
        self.crop_size, self.image_size, self.cropping_network = crop_size, image_size, cropping_network

    def discretize(self, data):
        """processor.py:162-176: data [T,1,H,W] -> [T,H,W] int64"""
        from . import ops as OO
        out_list = []
        for i in range(len(data)):
            cur = data[i][None]
            if torch.count_nonzero(cur) == 0:
                soft = torch.zeros_like(cur)
            else:
                soft = torch.softmax(self.cropping_network(OO.normalize_intensity(cur))["pred"], dim=1)
            out_list.append(torch.argmax(soft, dim=1).squeeze(0))
        return torch.stack(out_list, dim=0)

    def get_mean_centroid(self, data):
        """processor.py:140-160"""
        T, H, W = data.shape
        data = data.clone()
        data[data > 0] = 1
        cen = []
        for t in range(len(data)):
            cur = data[t]
            if torch.count_nonzero(cur) == 0:
                c = torch.tensor([H / 2, W / 2]).view(1, 2)
            else:
                coords = masks_to_boxes(cur.unsqueeze(0))
                x = coords[:, 0] + ((coords[:, 2] - coords[:, 0]) / 2)
                y = coords[:, 1] + ((coords[:, 3] - coords[:, 1]) / 2)
                c = torch.stack([x, y], dim=-1)
            cen.append(c)
        return torch.cat(cen, dim=0).mean(0).int()

    def preprocess_no_registration(self, data):
        """processor.py:232-237"""
        temp_volume = self.discretize(data)
        return self.get_mean_centroid(temp_volume), temp_volume

    def adjust_cropping_window(self, centroid):
        half = self.crop_size // 2
        x_low = max(0, int(centroid[0]) - half)
        x_high = min(self.image_size, int(centroid[0]) + half)
        y_low = max(0, int(centroid[1]) - half)
        y_high = min(self.image_size, int(centroid[1]) + half)
        if x_low == 0:
            x_high = self.crop_size
        if x_high == self.image_size:
            x_low = self.image_size - self.crop_size
        if y_low == 0:
            y_high = self.crop_size
        if y_high == self.image_size:
            y_low = self.image_size - self.crop_size
        return {"crop_indices": [x_low, x_high, y_low, y_high],
                "padding_need": torch.tensor([x_low, self.image_size - x_high, y_low, self.image_size - y_high])}

    def crop_and_pad(self, data, mean_centroid):
        p = self.adjust_cropping_window(mean_centroid)
        c = p["crop_indices"]
        vol = data[:, :, c[2]:c[3], c[0]:c[1]]
        assert vol.shape[-1] == self.crop_size
        return vol, p["padding_need"]

    def uncrop_no_registration(self, output, padding_need):
        assert len(output) == len(padding_need)
        return torch.stack([F.pad(output[b], pad=tuple(padding_need[b].tolist())) for b in range(len(output))], 0)


# ----------------------------------------------------------------------------- per-slice flow wrapper (row a21)
def pad_nd_image(image, new_shape, mode="constant", kwargs=None, return_slicer=False):
    """batchgenerators pad_nd_image (un-vendored, parity unpinned; call site SegFlowGaussian.py:3310): pad the trailing
    len(new_shape) axes to max(new, old), below = diff // 2, above = diff // 2 + diff % 2."""
    if kwargs is None:
        kwargs = {"constant_values": 0}
    old = np.array(image.shape[-len(new_shape):])
    new = np.array([max(n, o) for n, o in zip(new_shape, old)])
    diff = new - old
    below, above = diff // 2, diff // 2 + diff % 2
    pad_list = [[0, 0]] * (image.ndim - len(new_shape)) + [list(i) for i in zip(below, above)]
    res = np.pad(image, pad_list, mode, **kwargs) if diff.any() else image
    if not return_slicer:
        return res
    pad_arr = np.array(pad_list)
    pad_arr[:, 1] = np.array(res.shape) - pad_arr[:, 1]
    return res, [slice(*i) for i in pad_arr]


def predict_2d_tiled_flow(flow_net, seg_net, unlabeled, target, processor, mean_centroid, patch_size, do_mirroring=True, mirror_axes=(0, 1),
                          num_classes=4):
    """One slice through SegFlowGaussian._internal_predict_2D_2Dconv_tiled_flow (SegFlowGaussian.py:3294-3533) and
    _internal_maybe_mirror_and_pred_2D (:3075-3245), with the two substitutions the build documents: the
    segmentation comes from the 2-D U-Net under flip TTA (forward has no 'seg' output) and the heart centroid is an argument (the
    reference computes it with the cropping network, processor.py:232-237).

    unlabeled [T, 1, X, Y] (numpy, one slice over time), target [X, Y] label map of frame 0 or None (then frame 0's own argmax is
    propagated), mean_centroid (x, y) in the centre-cropped patch.  Returns (seg [T,X,Y], softmax [T,K,X,Y], flow [T,2,X,Y],
    registered [T,1,X,Y]) as float32 / int64 numpy arrays, cut back to the input size."""
    from . import ops as OO
    T = unlabeled.shape[0]
    data, slicer = pad_nd_image(unlabeled, patch_size, "constant", {"constant_values": 0}, True)                 # :3310
    H, W = data.shape[-2:]
    y1, y2 = int((H / 2) - (patch_size[0] / 2)), int((H / 2) + (patch_size[0] / 2))                             # :3391-3394
    x1, x2 = int((W / 2) - (patch_size[1] / 2)), int((W / 2) + (patch_size[1] / 2))
    x_in = torch.from_numpy(np.ascontiguousarray(data[:, :, y1:y2, x1:x2])).float()                             # [T,1,P,P]
    cropped, padding_need = processor.crop_and_pad(x_in, mean_centroid)                                          # :3101
    cropped = OO.normalize_intensity(cropped.clone())                                                            # :3108 (whole [T,1,h,w] block)
    cs = processor.crop_size
    with torch.no_grad():
        probs = mirror_and_predict_2d(seg_net, cropped, mirror_axes, do_mirroring)                               # [T,K,h,w]
        flow = torch.zeros(T, 2, cs, cs)
        idx = torch.arange(1, T)
        c1, c2 = (list(torch.chunk(idx, 2)) + [idx[:0]])[:2] if T > 1 else (idx, idx)
        for order in (torch.cat([torch.tensor([0]), c1]), torch.cat([torch.tensor([0]), torch.flip(c2, dims=[0])])):  # :3120-3127
            if len(order) > 1:
                bf = flow_net(cropped[order][:, None])["backward_flow"]                                          # [n-1,1,2,h,w]
                for j, t in enumerate(order[1:].tolist()):
                    flow[t] = bf[j, 0]
        if target is not None:
            tp = pad_nd_image(np.asarray(target)[None, None], patch_size, "constant", {"constant_values": 0}, False)
            lab = processor.crop_and_pad(torch.from_numpy(np.ascontiguousarray(tp[:, :, y1:y2, x1:x2])).float(), mean_centroid)[0]
        else:
            lab = probs[:1].argmax(1, keepdim=True).float()
        registered = OO.warp_labels(flow[:, None], lab)[:, 0].float()                                           # :3427 -> [T,1,h,w]
    pn = padding_need[None]

    def place(t):                                                                                                # :3450-3467 + slicer
        full = processor.uncrop_no_registration(t[None], pn)[0]
        canvas = torch.zeros(tuple(full.shape[:-2]) + (H, W), dtype=full.dtype)
        canvas[..., y1:y2, x1:x2] += full
        return canvas[..., slicer[-2], slicer[-1]].numpy()

    softmax, flow_f, reg_f = place(probs), place(flow), place(registered)
    return softmax.argmax(1), softmax, flow_f, reg_f
