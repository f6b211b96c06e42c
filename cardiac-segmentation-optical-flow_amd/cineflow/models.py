"""The flow / segmentation networks of the hot path, on the HIP C ABI.

Same class names, constructor keywords (the subset the YAML configs of the reference actually vary) and
``state_dict`` keys as the reference; `file:line` citations are relative to /root/reference.
"""
import copy

import numpy as np
import torch

from . import ops
from .nn import (Module, conv_norm, Conv2d, ConvTranspose2d, GroupNorm, Conv3d, ConvTranspose3d, InstanceNorm3d, ConvBlocks2DGroupLegacy, Encoder2D, Decoder2D, CrossAttentionLayer,
                 TransformerFlowEncoderSuccessiveNoEmb, ConvGRUCell, SpatialTransformer, VecInt)


# ------------------------------------------------------------------------------------------------ RAFT pieces
class CorrVolume(Module):
    """CorrVolume(radius, stride) -- nnunet/lib/raft.py is absent from the reference snapshot; call sites
    SegFlowGaussian.py:256-261, :1376-1377.  Spec: DESIGN.md "CorrVolume" (parity unpinned)."""

    def __init__(self, radius, stride):
        super().__init__()
        self.radius, self.stride = radius, stride

    def forward(self, cur, prev):
        return ops.corr_volume(cur, prev, self.radius, self.stride)


class CorrBlock:
    """CorrBlock(fmap1, fmap2, radius) -- nnunet/lib/raft_initial.py absent; published RAFT; call sites
    SegFlowGaussian.py:929, :935."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.pyramid = ops.corr_pyramid(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return ops.corr_lookup(self.pyramid, coords, self.num_levels, self.radius)


def coords_grid(batch, ht, wd, device):
    return ops.coords_grid(batch, ht, wd, device)


class BasicMotionEncoder(Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = Conv2d(256, 192, 3, padding=1)
        self.convf1 = Conv2d(2, 128, 7, padding=3)
        self.convf2 = Conv2d(128, 64, 3, padding=1)
        self.conv = Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr, out):
        """writes cat[relu(conv(cat[cor,flo])), flow] into channels [128,256) of `out` ([B,256,h,w] = cat[inp, motion])."""
        cor = self.convc2(self.convc1(corr, act="relu"), act="relu")
        flo = self.convf2(self.convf1(flow, act="relu"), act="relu")
        self.conv(cor, x2=flo, act="relu", out=out, out_coff=128)
        ops.copy_channels(flow, 0, 2, dst=out, dst_coff=128 + 126)
        return out


class SepConvGRU(Module):
    """Published RAFT SepConvGRU.  z and r gates of each pass are computed by ONE conv (weights concatenated on
    Cout at load time, order [r | z]) so the ConvGRU gating kernels are reused."""

    def __init__(self, hidden_dim=128, input_dim=256):
        super().__init__()
        c = hidden_dim + input_dim
        self.hidden_dim = hidden_dim
        self.convz1 = Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))

    def _prepare(self):
        # the fused [r | z] tensors are built lazily from the children's weights: a (re)load must not keep the previous checkpoint's
        for k in [k for k in self.__dict__ if k.startswith("_rz_")]:
            del self.__dict__[k]

    def _fused(self, cr, cz, c1):
        """[r | z] gate convolution of one pass as ONE layer (weights concatenated on Cout): packed for the f16-split kernel
        (split-aware for cat[h, x]) and, for CONV_MODE 'f32', as the fp32 kernel's transposed matrix."""
        if not hasattr(cr, "_wt"):
            raise RuntimeError("SepConvGRU weights not loaded")
        w = torch.cat([cr._p["weight"], cz._p["weight"]], dim=0).contiguous()
        c1k = c1 if c1 % ops.f16s_chunk(*cz.ks) else None
        return {"wt": ops.prep_conv_weight(w), "b": torch.cat([cr._p["bias"], cz._p["bias"]]).contiguous(),
                "pk": ops.pack_conv_weight_f16s(w, c1=c1k) if cz._f16s else None}

    def forward(self, h, x):
        H = self.hidden_dim
        for (cz, cr, cq) in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
            key = "_rz_%d_%d" % (id(cz), h.shape[1])
            if key not in self.__dict__:
                self.__dict__[key] = self._fused(cr, cz, h.shape[1])
            f = self.__dict__[key]
            if f["pk"] is not None and ops.CONV_MODE == "f16s" and ops.f16s_dynamic_ok(h, x, cz.ks[0]):
                gates = ops.conv2d_f16s(h, f["pk"][0], f["pk"][1], f["b"], 2 * H, cz.ks[0], cz.ks[1], 1, cz.pad, x2=x, act="sigmoid")  # [r | z]
            else:
                gates = ops.conv2d(h, f["wt"], f["b"], 2 * H, cz.ks[0], cz.ks[1], 1, cz.pad, x2=x, act="sigmoid")
            rh = ops.gru_reset_mul(gates, h)
            q = cq(rh, x2=x, act="tanh")
            h = ops.gru_blend(gates, h, q)
        return h


class FlowHead(Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = Conv2d(hidden_dim, 2, 3, padding=1)

    def forward(self, x, res=None):
        return self.conv2(self.conv1(x, act="relu"), res=res)


class BasicUpdateBlock(Module):
    """Published RAFT BasicUpdateBlock; call site SegFlowGaussian.py:942: net, up_mask, delta = update_block(net, inp, corr, flow)."""

    def __init__(self, hidden_dim=128, corr_levels=4, corr_radius=4):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = {0: Conv2d(128, 256, 3, padding=1), 2: Conv2d(256, 64 * 9, 1, padding=0)}

    def _prepare(self):
        self.__dict__.pop("_mask_scaled", None)      # derived from mask.2.bias on first use: dropped on every (re)load

    def forward(self, net, inp_motion, corr, flow):
        """inp_motion: [B,256,h,w] buffer whose first 128 channels hold `inp`; the motion features are written
        into the second half (the reference's torch.cat([inp, motion_features]))."""
        self.encoder(flow, corr, inp_motion)
        net = self.gru(net, inp_motion)
        delta = self.flow_head(net)
        m = self.mask[0](net, act="relu")
        c2 = self.mask[2]
        if "_mask_scaled" not in self.__dict__:
            self.__dict__["_mask_scaled"] = (0.25 * c2._p["bias"]).contiguous()  # 0.25 * (W x + b) = alpha * conv + 0.25 b
        if c2._f16s and ops.CONV_MODE == "f16s" and ops.f16s_dynamic_ok(m, None, 1):
            mask = ops.conv2d_f16s(m, c2._wpk, c2._ws, self.__dict__["_mask_scaled"], 576, 1, 1, alpha=0.25)
        else:
            mask = ops.conv2d(m, c2._wt, self.__dict__["_mask_scaled"], 576, 1, 1, alpha=0.25)
        return net, mask, delta


# ------------------------------------------------------------------------------------------------ SegFlowGaussian
class SegFlowGaussian(Module):
    """nnunet/network_architecture/SegFlowGaussian.py:70-357, built as nnunet/lib/training_utils.py:1460-1537 maps
    raft_config.yaml (motion_appearance=True, dim_feedforward=3072) or video.yaml (motion_appearance=False,
    dim_feedforward=2048).  `forward(x)` with x [T,B,1,H,W] returns {'backward_flow': [T-1,B,2,H,W]} (cumulative
    ED->t flow) exactly like :1330-1447 / :1813-1912; raft=True runs the RAFT loop of :875-969."""

    def __init__(self, image_size, in_dims=(6, 128, 256), out_encoder_dims=(64, 128, 256), d_model=256, conv_depth=(1, 1, 1),
                 skip_co_depth=(1, 1, 1), bottleneck_heads=4, nb_layers=1, dim_feedforward=3072, motion_appearance=True,
                 radius=(4, 4, 4, 4), stride=(4, 2, 1, 1), nb_conv=2, residual=True, extra_block=True, downsample_conv=2, raft=False,
                 raft_iters=12):
        super().__init__()
        in_dims, out_encoder_dims, conv_depth = list(in_dims), list(out_encoder_dims), list(conv_depth)
        self.num_stages = len(conv_depth)
        self.d_model, self.image_size = d_model, image_size
        self.motion_appearance, self.raft, self.raft_iters = motion_appearance, raft, raft_iters
        self.H = self.W = int(image_size / 2 ** self.num_stages)
        self.num_classes = 4

        self.integration = VecInt((image_size, image_size), 7)
        self.motion_estimation = SpatialTransformer((image_size, image_size))
        in_past = copy.copy(in_dims)
        in_past[0] = 6
        enc = dict(d_model=d_model, out_dims=out_encoder_dims, conv_depth=conv_depth, nb_conv=nb_conv, residual=residual,
                   downsample_conv=downsample_conv)
        self.memory_encoder = Encoder2D(in_dims=in_past, extra_block=extra_block, **enc)
        in_q = copy.copy(in_dims)
        self.skip_co_reduction_list = []
        if not motion_appearance:
            in_q[0] = 1
            self.query_encoder = Encoder2D(in_dims=in_q, extra_block=extra_block, **enc)
            self.cost_volume_encoder_list, self.cost_volume_computation_list = [], []
            for idx, (dim, nb) in enumerate(zip(out_encoder_dims, skip_co_depth)):
                self.cost_volume_computation_list.append(CorrVolume(radius=radius[idx], stride=stride[idx]))
                self.cost_volume_encoder_list.append(
                    ConvBlocks2DGroupLegacy((2 * radius[idx] + 1) ** 2, dim, 1, residual=residual))
                self.skip_co_reduction_list.append(ConvBlocks2DGroupLegacy(2 * dim, dim, nb, residual=residual))
        else:
            in_q[0] = 2
            self.query_encoder = Encoder2D(in_dims=in_q, extra_block=False, motion_appearance=True, **enc)
            for dim, nb in zip(out_encoder_dims, skip_co_depth):
                self.skip_co_reduction_list.append(ConvBlocks2DGroupLegacy(2 * dim, dim, nb, residual=residual))
        dec_in = in_dims[:]
        dec_in[0] = 4
        self.flow_decoder = Decoder2D(d_model=d_model, dot_multiplier=2, conv_depth=conv_depth[::-1], in_encoder_dims=dec_in[::-1],
                                      out_encoder_dims=out_encoder_dims[::-1], num_classes=2, nb_conv=nb_conv, residual=residual)
        self.gru_cell = ConvGRUCell(input_size=(self.H, self.W), input_dim=d_model, hidden_dim=d_model)
        self.reduce_transformer = ConvBlocks2DGroupLegacy(d_model * 2, d_model, 1, residual=residual)
        self.bottleneck1 = CrossAttentionLayer(d_model, bottleneck_heads, nb_layers, dim_feedforward)
        self.bottleneck2 = CrossAttentionLayer(d_model, bottleneck_heads, nb_layers, dim_feedforward)
        if raft:
            self.update_block = BasicUpdateBlock(hidden_dim=d_model // 2)

    def forward(self, x, keep_from=None, keep=None):
        """keep_from / keep (not in the reference, whose forward runs one sequence batch): from recurrence step `keep_from` on only the first
        `keep` sequences of the batch go on (their state is sliced out) -- lets two sequence groups of lengths T and T-1 share the launches of
        their common steps (cineflow.inference.predict_cine_slices).  The returned flow is [T-1, B, ...]; rows >= keep of the steps >= keep_from
        are zero.  No kernel mixes batch entries; per-sequence results differ from separate calls only through the launch shapes the batch size
        selects (~5e-6 px measured)."""
        if self.raft:
            assert keep_from is None, "ragged batches are built for the two recurrent dispatches only"
            return self.forward_multi_task_flow_deformable_raft(x)
        if self.motion_appearance:
            return self.forward_motion_appearance(x, keep_from, keep)
        return self.forward_multi_task_flow_deformable_cost_volume_transformer_cat(x, keep_from, keep)

    @staticmethod
    def _narrow(t, n):
        if isinstance(t, (list, tuple)):
            return [SegFlowGaussian._narrow(u, n) for u in t]
        return t[:n].contiguous()

    @staticmethod
    def _stack_flows(flows, B):
        """per-step cumulative flows (the last ones possibly for fewer sequences) -> [T-1, B, 2, H, W]"""
        if all(f.shape[0] == B for f in flows):
            return torch.stack(flows, dim=0)
        out = torch.zeros((len(flows), B) + tuple(flows[0].shape[1:]), dtype=flows[0].dtype, device=flows[0].device)
        for i, f in enumerate(flows):
            out[i, :f.shape[0]] = f
        return out

    def _step_tail(self, f1, f2, hidden, new_skips, cum, x0, xt):
        """SegFlowGaussian.py:1410-1435 == :1878-1905."""
        gru_in = self.reduce_transformer(f1, x2=f2)
        hidden = self.gru_cell(gru_in, hidden)
        flow = self.flow_decoder(hidden, new_skips)
        cum = ops.add(cum, flow)
        past_motion, past_skips = self.memory_encoder(ops.memory_input(x0, xt, cum))
        return hidden, cum, past_motion, past_skips

    def forward_motion_appearance(self, x, keep_from=None, keep=None):
        """SegFlowGaussian.py:1813-1912."""
        T, B, C, H, W = x.shape
        B_all = B
        dev = x.device
        cum = torch.zeros((B, 2, H, W), dtype=torch.float32, device=dev)
        hidden = torch.zeros((B, self.d_model, self.H, self.W), dtype=torch.float32, device=dev)
        past_motion, past_skips = self.memory_encoder(ops.memory_input(x[0], x[0], cum))
        pair = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        ops.copy_channels(x[0], 0, 1, dst=pair, dst_coff=0)
        ops.copy_channels(x[0], 0, 1, dst=pair, dst_coff=1)
        first_app, _, _ = self.query_encoder(pair)
        prev_app = first_app
        flows = []
        x0 = x[0]
        for t in range(1, T):
            xt, xp = x[t], x[t - 1]
            if keep_from is not None and t >= keep_from:
                if B != keep:      # the shorter group is done: slice the recurrent state of the longer one out of the batch
                    B = keep
                    cum, hidden, past_motion, past_skips, first_app, prev_app, x0 = self._narrow(
                        (cum, hidden, past_motion, past_skips, first_app, prev_app, x0), B)
                xt, xp = xt[:B], xp[:B]
            pair = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
            ops.copy_channels(xt, 0, 1, dst=pair, dst_coff=0)
            ops.copy_channels(xp, 0, 1, dst=pair, dst_coff=1)
            cur_app, _cur_motion, skips = self.query_encoder(pair)
            new_skips = [self.skip_co_reduction_list[s](skips[s], x2=past_skips[s]) for s in range(self.num_stages)]
            f1 = self.bottleneck1(query=cur_app, key=prev_app, value=prev_app)
            f2 = self.bottleneck2(query=cur_app, key=first_app, value=past_motion)
            hidden, cum, past_motion, past_skips = self._step_tail(f1, f2, hidden, new_skips, cum, x0, xt)
            flows.append(cum)
            prev_app = cur_app
        return {"backward_flow": self._stack_flows(flows, B_all)}

    def forward_multi_task_flow_deformable_cost_volume_transformer_cat(self, x, keep_from=None, keep=None):
        """SegFlowGaussian.py:1330-1447 (skip_co_type 'both', correlation_value False, warp False)."""
        T, B, C, H, W = x.shape
        B_all = B
        dev = x.device
        cum = torch.zeros((B, 2, H, W), dtype=torch.float32, device=dev)
        hidden = torch.zeros((B, self.d_model, self.H, self.W), dtype=torch.float32, device=dev)
        past_motion, past_skips = self.memory_encoder(ops.memory_input(x[0], x[0], cum))
        first_feat, first_skips = self.query_encoder(x[0])
        prev_feat, prev_skips = first_feat, first_skips
        flows = []
        x0 = x[0]
        for t in range(1, T):
            xt = x[t]
            if keep_from is not None and t >= keep_from:
                if B != keep:      # the shorter group is done: slice the recurrent state of the longer one out of the batch
                    B = keep
                    cum, hidden, past_motion, past_skips, first_feat, prev_feat, prev_skips, x0 = self._narrow(
                        (cum, hidden, past_motion, past_skips, first_feat, prev_feat, prev_skips, x0), B)
                xt = xt[:B].contiguous()
            cur_feat, cur_skips = self.query_encoder(xt)
            new_skips = []
            for s in range(self.num_stages):
                corr = self.cost_volume_computation_list[s](cur_skips[s], prev_skips[s])
                corr = self.cost_volume_encoder_list[s](corr)
                new_skips.append(self.skip_co_reduction_list[s](corr, x2=past_skips[s]))
            f1 = self.bottleneck1(query=cur_feat, key=prev_feat, value=prev_feat)
            f2 = self.bottleneck2(query=cur_feat, key=first_feat, value=past_motion)
            hidden, cum, past_motion, past_skips = self._step_tail(f1, f2, hidden, new_skips, cum, x0, xt)
            flows.append(cum)
            prev_feat, prev_skips = cur_feat, cur_skips
        return {"backward_flow": self._stack_flows(flows, B_all)}

    def forward_multi_task_flow_deformable_raft(self, x):
        """SegFlowGaussian.py:875-969; element [0] of the encoders' (feature, skips) tuple is used where the
        reference splits the tuple itself, and update_block is the published RAFT block."""
        T, B, C, H, W = x.shape
        dev = x.device
        half = self.d_model // 2
        flow_up = torch.zeros((B, 2, H, W), dtype=torch.float32, device=dev)
        cnet = self.memory_encoder(ops.memory_input(x[0], x[0], flow_up))[0]
        net = ops.copy_channels(cnet, 0, half, act="tanh")
        inp_motion = torch.empty((B, 2 * half, H // 8, W // 8), dtype=torch.float32, device=dev)
        ops.copy_channels(cnet, half, half, dst=inp_motion, dst_coff=0, act="relu")
        coords0 = coords_grid(B, H // 8, W // 8, dev)
        coords1 = coords_grid(B, H // 8, W // 8, dev)
        f1 = self.query_encoder(x[0])[0]
        out = []
        for t in range(1, T):
            f2 = self.query_encoder(x[t])[0]
            corr_fn = CorrBlock(f1, f2, radius=4)
            its = []
            for _ in range(self.raft_iters):
                corr = corr_fn(coords1)
                flow = ops.sub(coords1, coords0)
                net, up_mask, delta = self.update_block(net, inp_motion, corr, flow)
                coords1 = ops.add(coords1, delta)
                flow_up = ops.convex_upsample(ops.sub(coords1, coords0), up_mask)
                its.append(flow_up)
            out.append(torch.stack(its, dim=0))
            cnet = self.memory_encoder(ops.memory_input(x[0], x[t], flow_up))[0]
            ops.copy_channels(cnet, half, half, dst=inp_motion, dst_coff=0, act="relu")
        return {"backward_flow": torch.stack(out, dim=1)}

    def warp_linear(self, flow, labels):
        """SegFlowGaussian.py:3571-3580 (one_hot -> warp -> argmax), fused.  flow [T,B,2,H,W]; labels uint8 [B,H,W]
        -> uint8 [T,B,H,W]."""
        return ops.warp_labels(flow, labels, self.num_classes)


# ------------------------------------------------------------------------------------------------ successive model
class OpticalFlowModelSuccessive(Module):
    """nnunet/network_architecture/Optical_flow_model_successive.py:186-404 with successive.yaml values."""

    def __init__(self, image_size, nb_channels, in_dims=(6, 128, 256), out_encoder_dims=(64, 128, 256), conv_depth=(1, 1, 1),
                 bottleneck_heads=8, nb_layers=1, nb_conv=2, downsample_conv=1):
        super().__init__()
        in_dims, out_encoder_dims, conv_depth = list(in_dims), list(out_encoder_dims), list(conv_depth)
        self.num_stages = len(conv_depth)
        self.d_model = out_encoder_dims[-1] * 2
        self.image_size = image_size
        self.integration = VecInt((image_size, image_size), 7)
        in_dims[0] = nb_channels
        self.encoder = Encoder2D(d_model=self.d_model, out_dims=out_encoder_dims, in_dims=in_dims, conv_depth=conv_depth, nb_conv=nb_conv,
                                 extra_block=False, residual=False, downsample_conv=downsample_conv)
        dec_in = in_dims[:]
        dec_in[0] = 4
        self.flow_decoder = Decoder2D(d_model=self.d_model, dot_multiplier=2, conv_depth=conv_depth[::-1], in_encoder_dims=dec_in[::-1],
                                      out_encoder_dims=out_encoder_dims[::-1], num_classes=2, nb_conv=nb_conv, residual=False)
        self.bottleneck = TransformerFlowEncoderSuccessiveNoEmb(self.d_model, bottleneck_heads, nb_layers)
        self.skip_co_reduction_list = [ConvBlocks2DGroupLegacy(2 * d, d, 1, nb_conv=nb_conv) for d in out_encoder_dims]

    def forward(self, unlabeled, inference=False):
        T, B = unlabeled.shape[0], unlabeled.shape[1]
        # every frame through the shared encoder in ONE batched pass (frames are independent there)
        x = unlabeled.reshape((T * B,) + tuple(unlabeled.shape[2:]))
        feat, skips = self.encoder(x)
        feats = feat.view((T, B) + tuple(feat.shape[1:]))
        fwd = self.bottleneck(feats)  # [T-1,B,C,h,w]
        sk_t = [s.view((T, B) + tuple(s.shape[1:])) for s in skips]
        # adjacent-pair skip reductions and decoding, batched over the T-1 pairs
        n = (T - 1) * B
        red = [self.skip_co_reduction_list[s](sk_t[s][:-1].reshape((n,) + tuple(sk_t[s].shape[2:])),
                                              x2=sk_t[s][1:].reshape((n,) + tuple(sk_t[s].shape[2:])))
               for s in range(self.num_stages)]
        flow = self.flow_decoder(fwd.reshape((n,) + tuple(fwd.shape[2:])), red)
        if inference:
            flow = self.integration(flow)
        return {"flow": flow.view((T - 1, B) + tuple(flow.shape[1:]))}


class ModelWrap(Module):
    """Optical_flow_model_successive.py:58-134 (forward_from_ed, no_error=False)."""

    def __init__(self, model1, model2):
        super().__init__()
        self.model1, self.model2 = model1, model2
        self.image_size = model1.image_size
        self.motion_estimation = SpatialTransformer((self.image_size, self.image_size))

    def forward(self, x, inference=False):
        out2 = {}
        out1 = self.model1(x)
        if len(x) == 2:
            out2["flow"] = out1["flow"][0]
            return out1, out2
        flow1 = out1["flow"]
        B, _, H, W = x[0].shape
        cum = flow1[0]
        cums = [cum]
        for t in range(1, len(flow1)):
            reg1 = self.motion_estimation(flow=cum, original=x[t])
            reg2 = self.motion_estimation(flow=flow1[t], original=x[t + 1])
            xin = torch.empty((2, B, 6, H, W), dtype=torch.float32, device=x.device)
            for slot, (fl, a, b, reg) in enumerate(((cum, x[t], x[0], reg1), (flow1[t], x[t + 1], x[t], reg2))):
                ops.copy_channels(fl, 0, 2, dst=xin[slot], dst_coff=0)
                ops.copy_channels(a, 0, 1, dst=xin[slot], dst_coff=2)
                ops.copy_channels(b, 0, 1, dst=xin[slot], dst_coff=3)
                ops.copy_channels(reg, 0, 1, dst=xin[slot], dst_coff=4)
                ops.copy_channels(ops.sub(b, reg), 0, 1, dst=xin[slot], dst_coff=5)
            out = self.model2(xin, inference=inference)
            cum = ops.add(cum, out["flow"][0])
            cums.append(cum)
        out2["flow"] = cum
        out2["cumulated"] = torch.stack(cums, dim=0)
        return out1, out2


# ------------------------------------------------------------------------------------------------ Generic_UNet (2D)
class ConvDropoutNormNonlin(Module):
    """nnunet/network_architecture/generic_UNet.py:26-69: conv3x3 -> InstanceNorm(affine) -> LeakyReLU(0.01)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True)
        self.instnorm = GroupNorm(cout, cout)

    def forward(self, x, x2=None):
        return conv_norm(self.conv, self.instnorm, x, x2=x2, act="lrelu")


class StackedConvLayers(Module):
    """generic_UNet.py:79-144."""

    def __init__(self, cin, cout, num_convs, first_stride=None):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = [ConvDropoutNormNonlin(cin, cout, first_stride if first_stride is not None else 1)] + \
                      [ConvDropoutNormNonlin(cout, cout) for _ in range(num_convs - 1)]

    def forward(self, x, x2=None, pending=None, defer_last=None):
        """A block whose output feeds only the next convolution leaves its InstanceNorm + LeakyReLU to that convolution (applied while the
        tile is staged, ops.conv2d_f16s_prenorm): one 8-byte-per-element pass less per pair.
        pending = (raw conv output, its statistics, its norm module) handed over by the PREVIOUS stack (x is then ignored);
        defer_last = the convolution that will consume this stack's output alone: when it qualifies, the last block's norm is handed on
        as the returned `pending` instead of being applied.  Returns x, or (x, pending) when defer_last is given."""
        for i, b in enumerate(self.blocks):
            last = i == len(self.blocks) - 1
            if pending is not None:
                raw, ws, norm = pending
                B, C, H, W = raw.shape
                coef = ops.group_norm_coef(ws, norm._p["weight"], norm._p["bias"], norm.groups, B, C, H * W, norm.eps)
                y, ws_b = b.conv.prenorm(raw, coef, 0.01, stats_groups=b.instnorm.groups)
            else:
                kw = {} if (x2 is None or i > 0) else {"x2": x2}
                y, ws_b = b.conv(x, stats_groups=b.instnorm.groups, **kw)
            nxt = defer_last if last else self.blocks[i + 1].conv
            if nxt is not None and ws_b is not None and ops.CONV_MODE == "f16s" and nxt.ks == (3, 3) and nxt.prenorm_ok(y):
                pending = (y, ws_b, b.instnorm)
                continue
            if last and nxt is not None and ws_b is not None and nxt.ks == (1, 1) and nxt.stride == 1 and ops.norm_head_ok(y, nxt.cout):
                pending = (y, ws_b, b.instnorm)      # a 1x1 head takes the norm + LeakyReLU itself (ops.norm_head_1x1)
                continue
            pending = None
            x = b.instnorm(y, act="lrelu", ws=ws_b)
        return (x, pending) if defer_last is not None else x

    def first_conv(self):
        return self.blocks[0].conv


class Generic_UNet(Module):
    """generic_UNet.py:167-408 as nnUNetTrainerV2.py:147-169 builds it for 2-D (InstanceNorm affine, LeakyReLU,
    convolutional pooling / transposed-conv upsampling without bias, 1x1 heads without bias).  forward returns the
    full-resolution logits (deep supervision off at inference)."""

    MAX_FILTERS_2D = 480

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2, pool_op_kernel_sizes=None):
        """pool_op_kernel_sizes: the plans' per-stage pooling kernels (generic_UNet.py:247-248; strided first convolutions :283-285,
        transposed convolutions :343-344), e.g. [[2,2]]*5 + [[2,1]] for the ACDC 2-D patch (256, 224); default (2, 2) everywhere."""
        super().__init__()
        self.num_classes = num_classes
        self.input_channels = input_channels
        pool = [tuple(int(v) for v in p_) for p_ in (pool_op_kernel_sizes or [(2, 2)] * num_pool)]
        assert len(pool) == num_pool and all(p_ in ((2, 2), (2, 1), (1, 2)) for p_ in pool), "pooling kernels (2,2), (2,1) or (1,2), one per stage"
        self.pool_op_kernel_sizes = pool
        ctx, loc, tu, seg = [], [], [], []
        out_f, in_f = base_num_features, input_channels
        for d in range(num_pool):
            ctx.append(StackedConvLayers(in_f, out_f, num_conv_per_stage, pool[d - 1] if d != 0 else None))
            in_f = out_f
            out_f = min(int(np.round(out_f * 2)), self.MAX_FILTERS_2D)
        final = out_f
        ctx.append({0: StackedConvLayers(in_f, out_f, num_conv_per_stage - 1, pool[-1]), 1: StackedConvLayers(out_f, final, 1)})
        skip_ch = [c.output_channels for c in ctx[:-1]]
        for u in range(num_pool):
            from_down = final
            from_skip = skip_ch[-(1 + u)]
            final = from_skip
            tu.append(ConvTranspose2d(from_down, from_skip, bias=False, kernel_size=pool[-(u + 1)]))
            loc.append({0: StackedConvLayers(from_skip * 2, from_skip, num_conv_per_stage - 1), 1: StackedConvLayers(from_skip, final, 1)})
            seg.append(Conv2d(final, num_classes, 1, bias=False))
        self.conv_blocks_localization = loc
        self.conv_blocks_context = ctx
        self.tu = tu
        self.seg_outputs = seg

    def _children(self):
        # lists whose elements are {index: Module} dicts (nn.Sequential inside nn.ModuleList)
        for k in ("conv_blocks_localization", "conv_blocks_context"):
            for i, m in enumerate(getattr(self, k)):
                if isinstance(m, dict):
                    for j, mm in m.items():
                        yield "%s.%d.%d" % (k, i, j), mm
                else:
                    yield "%s.%d" % (k, i), m
        for k in ("tu", "seg_outputs"):
            for i, m in enumerate(getattr(self, k)):
                yield "%s.%d" % (k, i), m

    def forward(self, x):
        skips = []
        for d in range(len(self.conv_blocks_context) - 1):
            x = self.conv_blocks_context[d](x)
            skips.append(x)
        # the two stacks of the bottleneck and of every decoder stage are one chain: the first stack's norm rides into the second stack's
        # convolution instead of running as a pass of its own
        bott = self.conv_blocks_context[-1]
        x, pend = bott[0](x, defer_last=bott[1].first_conv())
        x = bott[1](x, pending=pend)
        head = self.seg_outputs[-1]
        for u in range(len(self.tu)):
            up = self.tu[u](x)
            blk = self.conv_blocks_localization[u]
            x, pend = blk[0](up, x2=skips[-(u + 1)], defer_last=blk[1].first_conv())
            if u == len(self.tu) - 1:
                # the last stack hands its norm to the 1x1 head: one pass over the raw map instead of apply + 1x1 convolution
                x, pend = blk[1](x, pending=pend, defer_last=head)
                if pend is not None:
                    raw, ws, norm = pend
                    B, C, H, W = raw.shape
                    coef = ops.group_norm_coef(ws, norm._p["weight"], norm._p["bias"], norm.groups, B, C, H * W, norm.eps)
                    return ops.norm_head_1x1(raw, coef, 0.01, head._p["weight"], head._p.get("bias"))
            else:
                x = blk[1](x, pending=pend)
        return head(x)


# ------------------------------------------------------------------------------------------------ Generic_UNet (3D)
class ConvDropoutNormNonlin3D(Module):
    """generic_UNet.py:26-69 with conv_op = nn.Conv3d: conv -> InstanceNorm3d(affine) -> LeakyReLU(0.01)."""

    def __init__(self, cin, cout, kernel=(3, 3, 3), stride=(1, 1, 1)):
        super().__init__()
        self.conv = Conv3d(cin, cout, kernel, stride, bias=True)
        self.instnorm = InstanceNorm3d(cout)

    def forward(self, x, x2=None):
        return self.instnorm(self.conv(x, x2=x2), act="lrelu")


class StackedConvLayers3D(Module):
    """generic_UNet.py:79-144."""

    def __init__(self, cin, cout, num_convs, kernel, first_stride=None):
        super().__init__()
        self.input_channels, self.output_channels = cin, cout
        self.blocks = [ConvDropoutNormNonlin3D(cin, cout, kernel, first_stride if first_stride is not None else (1, 1, 1))] + \
                      [ConvDropoutNormNonlin3D(cout, cout, kernel) for _ in range(num_convs - 1)]

    def forward(self, x, x2=None, pending=None, defer_last=None):
        for i, b in enumerate(self.blocks):
            x = b(x, x2=x2) if i == 0 else b(x)
        return (x, None) if defer_last is not None else x        # (3-D stacks apply every norm themselves)

    def first_conv(self):
        return self.blocks[0].conv


class Generic_UNet3D(Generic_UNet):
    """generic_UNet.py:167-408 with conv_op = nn.Conv3d -- the network `_internal_predict_3D_3Dconv_tiled` drives
    (neural_network.py:292-430): per-stage pool_op_kernel_sizes / conv_kernel_sizes from the plans (anisotropic (1,2,2) /
    (1,3,3) stages allowed), convolutional pooling and upsampling, MAX_NUM_FILTERS_3D = 320.  Same state_dict keys as the
    reference.  forward: [B,C,D,H,W] -> full-resolution logits [B,K,D,H,W]."""

    MAX_NUM_FILTERS_3D = 320

    def __init__(self, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2, pool_op_kernel_sizes=None,
                 conv_kernel_sizes=None):
        Module.__init__(self)
        self.num_classes = num_classes
        self.input_channels = input_channels
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
        ctx.append({0: StackedConvLayers3D(in_f, out_f, num_conv_per_stage - 1, kern[num_pool], pool[-1]),
                    1: StackedConvLayers3D(out_f, final, 1, kern[num_pool])})
        skip_ch = [c.output_channels for c in ctx[:-1]]
        for u in range(num_pool):
            from_down = final
            from_skip = skip_ch[-(1 + u)]
            final = from_skip
            tu.append(ConvTranspose3d(from_down, from_skip, pool[-(u + 1)], bias=False))
            loc.append({0: StackedConvLayers3D(from_skip * 2, from_skip, num_conv_per_stage - 1, kern[-(u + 1)]),
                        1: StackedConvLayers3D(from_skip, final, 1, kern[-(u + 1)])})
            seg.append(Conv3d(final, num_classes, (1, 1, 1), bias=False))
        self.conv_blocks_localization = loc
        self.conv_blocks_context = ctx
        self.tu = tu
        self.seg_outputs = seg
