"""Sliding-window / TTA inference drivers of the hot path.

Mirrors nnunet/network_architecture/neural_network.py (`SegmentationNetwork`) and the flow-specific overrides of
nnunet/network_architecture/SegFlowGaussian.py, with the per-tile host<->device round trips of the reference removed:
tiles, mirrored copies, Gaussian weighting, accumulation, argmax and the label warp all stay in HBM, and every tile
of every slice goes through the network as ONE batch.  `file:line` citations are relative to /root/reference.
"""
import os

import numpy as np
import torch

from . import ops


# ------------------------------------------------------------------------------------------------ host-side helpers
def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """SegmentationNetwork._compute_steps_for_sliding_window, neural_network.py:267-290 (pure host integer logic)."""
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be as large or larger than patch_size"
    assert 0 < step_size <= 1, "step_size must be larger than 0 and smaller or equal to 1"
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step_value = image_size[dim] - patch_size[dim]
        actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def get_gaussian(patch_size, sigma_scale=1.0 / 8):
    """SegmentationNetwork._get_gaussian, neural_network.py:251-264 (computed once per patch size, on the host)."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode="constant", cval=0)
    g = (g / np.max(g) * 1).astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_nd_image(image, new_shape, mode="constant", kwargs=None, return_slicer=False):
    """batchgenerators pad_nd_image semantics (call sites neural_network.py:644, SegFlowGaussian.py:3310): pad the
    trailing len(new_shape) axes to max(new, old), below = diff//2, above = diff//2 + diff%2."""
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


_gauss_cache = {}


def _gaussian_on(device, patch_size):
    key = (str(device), tuple(patch_size))
    if key not in _gauss_cache:
        _gauss_cache[key] = torch.from_numpy(get_gaussian(tuple(patch_size))).to(device)
    return _gauss_cache[key]


# ------------------------------------------------------------------------------------------------ TTA + tiles
def mirror_and_predict_2d(net, x, mirror_axes=(0, 1), do_mirroring=True, mult=None):
    """SegmentationNetwork._internal_maybe_mirror_and_pred_2D, neural_network.py:573-621 (upstream 4-argument
    semantics).  x [B,C,X,Y] on the GPU; returns the TTA-averaged softmax [B,K,X,Y] (times `mult` [X,Y])."""
    B, _, X, Y = x.shape
    acc = torch.zeros((B, net.num_classes, X, Y), dtype=torch.float32, device=x.device)
    n = 2 ** len(mirror_axes) if do_mirroring else 1
    variants = [(0, 0)]
    if do_mirroring:
        if 1 in mirror_axes:
            variants.append((0, 1))
        if 0 in mirror_axes:
            variants.append((1, 0))
        if 0 in mirror_axes and 1 in mirror_axes:
            variants.append((1, 1))
    for fh, fw in variants:
        xin = x if (fh, fw) == (0, 0) else ops.flip2d(x, fh, fw)
        ops.tta_accumulate(net(xin), acc, fh, fw, 1.0 / n)
    if mult is not None:
        ops.mul(acc, mult, out=acc)
    return acc


def predict_3D_2Dconv_tiled(net, x, patch_size, step_size=0.5, do_mirroring=True, mirror_axes=(0, 1), use_gaussian=True,
                            pad_border_mode="constant", pad_kwargs=None, max_batch=64, return_device=False):
    """SegmentationNetwork._internal_predict_3D_2Dconv_tiled (neural_network.py:814-857) over
    _internal_predict_2D_2Dconv_tiled (:623-769).  x: numpy [C,Z,X,Y] -> (seg [Z,X,Y] uint8, softmax [K,Z,X,Y] fp32).
    All Z slices and all tiles are batched through the network (the reference loops tile by tile)."""
    assert x.ndim == 4, "x must be (c, z, x, y)"
    C, Z = x.shape[0], x.shape[1]
    patch_size = tuple(patch_size)
    data, slicer = pad_nd_image(x, patch_size, pad_border_mode, pad_kwargs, True)
    Xp, Yp = data.shape[2], data.shape[3]
    steps = compute_steps_for_sliding_window(patch_size, (Xp, Yp), step_size)
    tiles = [(lx, ly) for lx in steps[0] for ly in steps[1]]
    dev = torch.device("cuda", torch.cuda.current_device())
    vol = torch.from_numpy(np.ascontiguousarray(data.transpose(1, 0, 2, 3))).to(dev, dtype=torch.float32)  # [Z,C,Xp,Yp]
    K = net.num_classes
    gauss = _gaussian_on(dev, patch_size) if (use_gaussian and len(tiles) > 1) else None
    agg = torch.zeros((Z, K, Xp, Yp), dtype=torch.float32, device=dev)
    cnt = torch.zeros((Z, K, Xp, Yp), dtype=torch.float32, device=dev)
    jobs = [(z, lx, ly) for z in range(Z) for (lx, ly) in tiles]
    for i0 in range(0, len(jobs), max_batch):
        chunk = jobs[i0:i0 + max_batch]
        batch = torch.empty((len(chunk), C) + patch_size, dtype=torch.float32, device=dev)
        for j, (z, lx, ly) in enumerate(chunk):
            batch[j] = ops.crop2d(vol[z], lx, ly, patch_size[0], patch_size[1])
        pred = mirror_and_predict_2d(net, batch, mirror_axes, do_mirroring, gauss)
        for j, (z, lx, ly) in enumerate(chunk):
            ops.tile_accumulate(pred[j], gauss, agg[z], cnt[z], lx, ly)
    segs, probs = [], []
    for z in range(Z):
        s, p = ops.tile_finalize(agg[z], cnt[z])
        segs.append(s)
        probs.append(p)
    seg = torch.stack(segs, 0)[:, slicer[2], slicer[3]]
    prob = torch.stack(probs, 1)[:, :, slicer[2], slicer[3]]
    if return_device:
        return seg, prob
    return seg.cpu().numpy(), prob.cpu().numpy()


def mirror_and_predict_3d(net, x, mirror_axes=(0, 1, 2), do_mirroring=True, mult=None):
    """SegmentationNetwork._internal_maybe_mirror_and_pred_3D, neural_network.py:506-571.  x [B,C,X,Y,Z] on the GPU; returns the
    TTA-averaged softmax [B,K,X,Y,Z] (times `mult` [X,Y,Z]); up to 8 flip combinations of the axes named in `mirror_axes`."""
    B = x.shape[0]
    acc = torch.zeros((B, net.num_classes) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    n = 2 ** len(mirror_axes) if do_mirroring else 1
    variants = [(0, 0, 0)]
    if do_mirroring:
        for m in range(1, 8):   # the reference's order: bit 0 = axis 2 (dim 4), bit 1 = axis 1, bit 2 = axis 0
            f = (bool(m & 4), bool(m & 2), bool(m & 1))
            if all((not f[a]) or (a in mirror_axes) for a in range(3)):
                variants.append(tuple(int(v) for v in f))
    for f in variants:
        xin = x if f == (0, 0, 0) else ops.flip3d(x, *f)
        ops.tta_accumulate_3d(net(xin), acc, f[0], f[1], f[2], 1.0 / n)
    if mult is not None:
        ops.mul(acc, mult, out=acc)
    return acc


def predict_3D_3Dconv_tiled(net, x, patch_size, step_size=0.5, do_mirroring=True, mirror_axes=(0, 1, 2), use_gaussian=True,
                            pad_border_mode="constant", pad_kwargs=None, return_device=False):
    """SegmentationNetwork._internal_predict_3D_3Dconv_tiled (neural_network.py:292-430).  x: numpy [C,X,Y,Z] ->
    (seg [X,Y,Z] uint8, softmax [K,X,Y,Z] fp32).  Volume, aggregation buffers and Gaussian stay on the GPU; one 3-D patch
    per network call as in the reference."""
    assert x.ndim == 4, "x must be (c, x, y, z)"
    patch_size = tuple(patch_size)
    data, slicer = pad_nd_image(x, patch_size, pad_border_mode, pad_kwargs, True)
    Xp, Yp, Zp = data.shape[1:]
    steps = compute_steps_for_sliding_window(patch_size, (Xp, Yp, Zp), step_size)
    tiles = [(lx, ly, lz) for lx in steps[0] for ly in steps[1] for lz in steps[2]]
    dev = torch.device("cuda", torch.cuda.current_device())
    vol = torch.from_numpy(np.ascontiguousarray(data)).to(dev, dtype=torch.float32)  # [C,Xp,Yp,Zp]
    K = net.num_classes
    gauss = _gaussian_on(dev, patch_size) if (use_gaussian and len(tiles) > 1) else None
    agg = torch.zeros((K, Xp, Yp, Zp), dtype=torch.float32, device=dev)
    cnt = torch.zeros((K, Xp, Yp, Zp), dtype=torch.float32, device=dev)
    px, py, pz = patch_size
    for (lx, ly, lz) in tiles:
        tile = vol[None, :, lx:lx + px, ly:ly + py, lz:lz + pz].contiguous()
        pred = mirror_and_predict_3d(net, tile, mirror_axes, do_mirroring, gauss)
        ops.tile_accumulate_3d(pred[0], gauss, agg, cnt, lx, ly, lz)
    seg, prob = ops.tile_finalize(agg.view(K, Xp, Yp * Zp), cnt.view(K, Xp, Yp * Zp))
    seg = seg.view(Xp, Yp, Zp)[slicer[1], slicer[2], slicer[3]]
    prob = prob.view(K, Xp, Yp, Zp)[:, slicer[1], slicer[2], slicer[3]]
    if return_device:
        return seg, prob
    return seg.cpu().numpy(), prob.cpu().numpy()


# ------------------------------------------------------------------------------------------------ Processor
class Processor:
    """nnunet/training/network_training/processor.py: the crop / un-crop arithmetic (:109-138, :178-186, :223-230) and the heart
    centroid (:140-160 get_mean_centroid, :162-176 discretize, :232-237 preprocess_no_registration).

    cropping_network: any callable mapping a device batch [N,1,H,W] to {'pred': logits [N,K,H,W]} (the reference builds a 2-class
    MTLmodel from adversarial_acdc.yaml, voxelmorph_saver_Lib.py:328-335; `CroppingNet` below wraps any cineflow network).  Without
    one the centroid must be passed by the caller (or defaults to the image centre)."""

    def __init__(self, crop_size, image_size, cropping_network=None):
        self.crop_size, self.image_size, self.cropping_network = crop_size, image_size, cropping_network

    # -- processor.py:162-176
    def discretize(self, data):
        """data [T,1,H,W] (device) -> label maps uint8 [T,H,W]: per frame NormalizeIntensity, network, softmax, argmax; an all-zero
        frame gives an all-zero map (the network is not consulted for it).  All frames go through the network as one batch."""
        T, one, H, W = data.shape
        assert one == 1
        x = data.contiguous().clone()
        nonempty = (ops.frame_boxes(x.view(T, H, W))[:, 0] >= 0)                  # frames with a non-zero pixel
        flat = x.view(T, 1, H * W)
        ops.group_norm(flat, None, None, 1, eps=0.0, out=flat)                    # per-frame z-score (population std), in place
        # a constant frame has std 0: monai's NormalizeIntensity then divides by 1 (-> all zeros), the kernel's 0 * inf is NaN.  Every
        # z-scored frame that is not finite is such a frame: it becomes zeros, which is also what keeps the all-zero frame out of the batch
        const = ~torch.isfinite(flat.view(T, -1)[:, 0])
        x[const | ~nonempty] = 0
        logits = self.cropping_network(x)["pred"]
        lab = ops.argmax_channels(logits.contiguous())                            # softmax is monotone: argmax of the logits
        lab[~nonempty] = 0
        return lab

    # -- processor.py:140-160
    def get_mean_centroid(self, data):
        """label maps [T,H,W] (device) -> (x, y) int: mean over the frames of the bounding-box centre of the non-zero pixels; a frame
        without any contributes (H/2, W/2) -- in that order, as the reference writes it."""
        T, H, W = data.shape
        boxes = ops.frame_boxes(data.contiguous()).cpu().to(torch.float32)       # [T,4] x1, y1, x2, y2 (tiny: the arithmetic below is the reference's)
        cen = []
        for t in range(T):
            if boxes[t, 0] < 0:
                cen.append(torch.tensor([H / 2, W / 2]).view(1, 2))
            else:
                x = boxes[t, 0] + ((boxes[t, 2] - boxes[t, 0]) / 2)
                y = boxes[t, 1] + ((boxes[t, 3] - boxes[t, 1]) / 2)
                cen.append(torch.stack([x, y], dim=-1).view(1, 2))
        return torch.cat(cen, dim=0).mean(0).int()

    # -- processor.py:232-237
    def preprocess_no_registration(self, data):
        """data [T,1,H,W] -> (mean_centroid (x, y) int tensor, label maps [T,H,W])"""
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
                "padding_need": [x_low, self.image_size - x_high, y_low, self.image_size - y_high]}

    def crop_and_pad(self, data, mean_centroid):
        """data [T,1,H,W] (GPU) -> ([T,1,crop,crop], padding_need [left,right,top,bottom])."""
        p = self.adjust_cropping_window(mean_centroid)
        c = p["crop_indices"]
        vol = ops.crop2d(data, c[2], c[0], c[3] - c[2], c[1] - c[0])
        assert vol.shape[-1] == self.crop_size, vol.shape
        return vol, p["padding_need"]

    def uncrop_no_registration(self, output, padding_need):
        """output [...,h,w] -> zero-padded [...,H,W] (F.pad(pad=(left,right,top,bottom)))."""
        left, right, top, bottom = (int(v) for v in padding_need)
        h, w = output.shape[-2:]
        return ops.pad2d(output, top, left, h + top + bottom, w + left + right)


class CroppingNet:
    """Adapter giving a cineflow network (e.g. Generic_UNet(1, base, 2, pools)) the interface Processor.discretize expects from the
    reference's cropping network: net(x)['pred'] = logits."""

    def __init__(self, net):
        self.net = net

    def __call__(self, x):
        return {"pred": self.net(x)}


# ------------------------------------------------------------------------------------------------ joint seg + flow
def normalize_intensity_(x):
    """monai NormalizeIntensity() defaults on a [T,1,h,w] block (SegFlowGaussian.py:3108): whole-block z-score with
    population std, done by the GroupNorm kernel with one group, eps 0 and no affine."""
    flat = x.view(1, 1, -1)
    ops.group_norm(flat, None, None, 1, eps=0.0, out=flat)
    return x


def chunk_orders(T):
    """SegFlowGaussian.py:3120-3127: frames 1..T-1 split in two chunks (torch.chunk), the second reversed, both
    starting at frame 0 (ED)."""
    idx = list(range(1, T))
    n1 = -(-len(idx) // 2) if len(idx) else 0
    c1, c2 = idx[:n1], idx[n1:]
    return [0] + c1, [0] + c2[::-1]


# CF_TWO_STREAMS=1: segmentation U-Net and flow recurrence on two HIP streams.  Measured +2.6 % frames/s (676.7 vs 659.2 on one box); OFF by
# default because overlapping kernels make the per-kernel event durations of bench.py's roofline (and rocprof's) meaningless.
TWO_STREAMS = os.environ.get("CF_TWO_STREAMS", "0") == "1"
# CF_RAGGED_CHUNKS=0: the two ED-anchored half sequences of unequal length run one after the other (A/B knob)
RAGGED_CHUNKS = os.environ.get("CF_RAGGED_CHUNKS", "1") == "1"
_side = {}


def _side_stream(dev):
    key = str(dev)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=dev)
    return _side[key]


def predict_cine_slices(flow_net, seg_net, frames, ed_labels=None, do_mirroring=True, mirror_axes=(0, 1), seg_mixed_precision=False):
    """The joint hot path of BASELINE.json config 4 for a batch of slices that are already cropped to the network's
    patch (SegFlowGaussian._internal_maybe_mirror_and_pred_2D :3120-3230 + _internal_predict_2D_2Dconv_tiled_flow
    :3427, with the segmentation coming from the 2-D U-Net because SegFlowGaussian.forward has no 'seg' output --
    SURVEY.md section 0.1).

    frames [T,B,1,H,W] float32 on the GPU (B = slices x patients), z-scored;  ed_labels uint8 [B,H,W] or None
    (None -> argmax of the ED frame's segmentation is propagated).
    seg_mixed_precision: the segmentation U-Net's convolutions run in the one-term product mode (ops.conv_terms(1): operands rounded to fp16,
    fp32 accumulation and norms) -- the reference's `mixed_precision=True` on the segmentation path (neural_network.py:140-146); the flow
    network always stays f32-class (SegFlowGaussian.py:2905-2909).
    Returns dict(seg uint8 [T,B,H,W], softmax [T,B,K,H,W], flow [T,B,2,H,W] (frame 0 zero), registered uint8 [T,B,H,W]).
    """
    T, B, _, H, W = frames.shape
    dev = frames.device
    seg_terms = 1 if seg_mixed_precision else 3
    # segmentation: every frame of every slice is independent -> one batch, flip-TTA on the softmax (:3165-3226).  It does not depend on the
    # flow recurrence, so it CAN be issued on a second HIP stream: the recurrence's small launches (attention, gates, 32x32 maps) leave CUs
    # idle that the U-Net's large launches fill (opt-in, see TWO_STREAMS)
    side = None
    if TWO_STREAMS:
        side = _side_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), ops.conv_terms(seg_terms):
            probs = mirror_and_predict_2d(seg_net, frames.reshape(T * B, 1, H, W), mirror_axes, do_mirroring)
            seg = ops.argmax_channels(probs).view(T, B, H, W)
    else:
        with ops.conv_terms(seg_terms):
            probs = mirror_and_predict_2d(seg_net, frames.reshape(T * B, 1, H, W), mirror_axes, do_mirroring)
        seg = ops.argmax_channels(probs).view(T, B, H, W)
    # flow: two half sequences that both start at ED, the second one backwards in time (:3120-3127); flow is not
    # TTA-averaged (:3162).  Both chunks run as one batch of 2B sequences when they have equal length.
    flow = torch.zeros((T, B, 2, H, W), dtype=torch.float32, device=dev)
    o1, o2 = chunk_orders(T)
    ragged = len(o1) == len(o2) + 1 and len(o2) > 1 and RAGGED_CHUNKS and hasattr(flow_net, "_narrow") and not getattr(flow_net, "raft", False)
    if len(o1) == len(o2) and len(o1) > 1:
        xin = torch.cat([frames[o1], frames[o2]], dim=1)  # [Tc, 2B, 1, H, W]  (copies only)
        bf = flow_net(xin)["backward_flow"]
        for j, t in enumerate(o1[1:]):
            flow[t] = bf[j, :B]
        for j, t in enumerate(o2[1:]):
            flow[t] = bf[j, B:]
    elif ragged:
        # T even (T = 30: 15 + 14 frames behind ED): the common steps of both groups run as one batch of 2B sequences, the last step of the
        # longer group alone.  Twice the work per launch for 28 of 29 steps; per-sequence numbers are those of separate calls.
        o2p = o2 + [o2[-1]]                                 # padding frame of the shorter group: never consumed
        xin = torch.cat([frames[o1], frames[o2p]], dim=1)   # [Tc, 2B, 1, H, W]
        bf = flow_net(xin, keep_from=len(o2), keep=B)["backward_flow"]
        for j, t in enumerate(o1[1:]):
            flow[t] = bf[j, :B]
        for j, t in enumerate(o2[1:]):
            flow[t] = bf[j, B:]
    else:
        for order in (o1, o2):
            if len(order) > 1:
                bf = flow_net(frames[order].contiguous())["backward_flow"]
                for j, t in enumerate(order[1:]):
                    flow[t] = bf[j]
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
        probs.record_stream(torch.cuda.current_stream(dev))
        seg.record_stream(torch.cuda.current_stream(dev))
    if ed_labels is None:
        ed_labels = seg[0].contiguous()
    registered = ops.warp_labels(flow, ed_labels, flow_net.num_classes if hasattr(flow_net, "num_classes") else 4)
    return {"seg": seg, "softmax": probs.view(T, B, -1, H, W), "flow": flow, "registered": registered}
