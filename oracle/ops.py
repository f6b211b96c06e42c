"""Oracle restatement of the small operators of the hot path (CPU, fp32).

Test infrastructure only -- see oracle/__init__.py.  All `file:line`
citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- warp
def identity_grid(shape, device="cpu"):
    """nnunet/network_architecture/integration.py:48-52 (meshgrid 'ij': channel 0 = row)."""
    vectors = [torch.arange(0, s, device=device) for s in shape]
    grids = torch.meshgrid(*vectors, indexing="ij")
    return torch.stack(grids).unsqueeze(0).float()


def warp_bilinear(flow, src, mode="bilinear"):
    """SpatialTransformer.forward, integration.py:61-79.

    flow [B,2,H,W] (channel 0 = row displacement), src [B,C,H,W].
    Same normalise -> grid_sample(align_corners=True, zeros) round trip as the
    reference, so the fp32 rounding path is identical.
    """
    shape = flow.shape[2:]
    new_locs = identity_grid(shape, flow.device) + flow
    for i in range(len(shape)):
        new_locs[:, i, ...] = 2 * (new_locs[:, i, ...] / (shape[i] - 1) - 0.5)
    if len(shape) == 2:
        new_locs = new_locs.permute(0, 2, 3, 1)[..., [1, 0]]
    else:
        new_locs = new_locs.permute(0, 2, 3, 4, 1)[..., [2, 1, 0]]
    return F.grid_sample(src, new_locs, align_corners=True, mode=mode)


def vecint(vec, nsteps=7):
    """VecInt.forward, integration.py:95-99: scaling and squaring."""
    vec = vec * (1.0 / (2 ** nsteps))
    for _ in range(nsteps):
        vec = vec + warp_bilinear(vec, vec)
    return vec


def warp_labels(flow, labels, num_classes=4):
    """warp_linear, nnunet/network_architecture/SegFlowGaussian.py:3571-3580.

    flow [T,B,2,H,W]; labels [B,1,H,W] (ED label map).  one_hot -> warp per
    frame -> argmax (first maximal index wins).  Returns [T,B,1,H,W] int64.
    """
    onehot = F.one_hot(labels[:, 0].long(), num_classes=num_classes).permute(0, 3, 1, 2).contiguous().float()
    out = []
    for t in range(flow.shape[0]):
        reg = warp_bilinear(flow[t], onehot)
        out.append(torch.argmax(reg, dim=1, keepdim=True))
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------- jacobian
def jacobian_determinant(disp):
    """nnunet/compute_jacobian.py:16-59 (2-D and 3-D).

    disp: numpy [*vol, nd]; pystrum.volsize2ndgrid == np.meshgrid(indexing='ij').
    """
    disp = np.asarray(disp)
    volshape = disp.shape[:-1]
    nb_dims = len(volshape)
    assert nb_dims in (2, 3)
    grid = np.stack(np.meshgrid(*[np.arange(s) for s in volshape], indexing="ij"), nb_dims)
    J = np.gradient(disp + grid)
    if nb_dims == 3:
        dx, dy, dz = J[0], J[1], J[2]
        d0 = dx[..., 0] * (dy[..., 1] * dz[..., 2] - dy[..., 2] * dz[..., 1])
        d1 = dx[..., 1] * (dy[..., 0] * dz[..., 2] - dy[..., 2] * dz[..., 0])
        d2 = dx[..., 2] * (dy[..., 0] * dz[..., 1] - dy[..., 1] * dz[..., 0])
        return d0 - d1 + d2
    dfdx, dfdy = J[0], J[1]
    return dfdx[..., 0] * dfdy[..., 1] - dfdy[..., 0] * dfdx[..., 1]


# --------------------------------------------------------------------------- correlation (PARITY UNPINNED)
def corr_volume(cur, prev, radius=4, stride=1):
    """CorrVolume(radius, stride)(cur, prev) -- source absent (nnunet/lib/raft.py).

    Spec fixed by the build from the call sites SegFlowGaussian.py:256-261,
    :1376-1377 and raft_config.yaml:43-44:
      out[b, (dy+r)*(2r+1)+(dx+r), y, x] = mean_c cur[b,c,y,x] * prev[b,c,y+dy*stride, x+dx*stride]
    with zeros outside the previous map.  -> [B,(2r+1)^2,H,W]
    """
    B, C, H, W = cur.shape
    r, s = radius, stride
    pad = r * s
    prev_p = F.pad(prev, (pad, pad, pad, pad))
    out = []
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            sh = prev_p[:, :, pad + dy * s: pad + dy * s + H, pad + dx * s: pad + dx * s + W]
            out.append((cur * sh).mean(dim=1))
    return torch.stack(out, dim=1)


def coords_grid(batch, ht, wd, device="cpu"):
    """coords_grid -- source absent (nnunet/lib/raft_initial.py); call site SegFlowGaussian.py:838-839.
    Published RAFT definition: channel 0 = x (column), channel 1 = y (row)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    coords = torch.stack([xs, ys], dim=0).float()
    return coords[None].repeat(batch, 1, 1, 1)


def corr_allpairs(fmap1, fmap2):
    """CorrBlock.corr (published RAFT): fmap1^T fmap2 / sqrt(C) -> [B,H*W,H,W]."""
    B, C, H, W = fmap1.shape
    f1 = fmap1.view(B, C, H * W)
    f2 = fmap2.view(B, C, H * W)
    corr = torch.matmul(f1.transpose(1, 2), f2)
    return (corr / math.sqrt(float(C))).view(B, H * W, H, W)


def corr_pyramid(corr, num_levels=4):
    """2x2 average-pool pyramid of the all-pairs volume. corr [B,N,H,W] -> list of [B,N,H/2^l,W/2^l]."""
    pyr = [corr]
    for _ in range(num_levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def _bilinear_sampler(img, coords):
    """published RAFT core/utils/utils.py bilinear_sampler: coords (...,2) = (x, y) in pixels."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    grid = torch.cat([xgrid, ygrid], dim=-1)
    return F.grid_sample(img, grid, align_corners=True)


def corr_lookup(pyramid, coords, radius=4):
    """CorrBlock.__call__(coords) (published RAFT core/corr.py), call site SegFlowGaussian.py:935.

    pyramid: list of [B,N,Hl,Wl] with N=H*W; coords [B,2,H,W] (x,y).  -> [B, L*(2r+1)^2, H, W].
    Channel (within a level) index i*(2r+1)+j samples at (x + (i-r), y + (j-r)) / 2^level
    (RAFT's meshgrid(dy, dx) ordering).
    """
    r = radius
    B, _, H, W = coords.shape
    coords = coords.permute(0, 2, 3, 1)
    out = []
    for i, corr in enumerate(pyramid):
        Hl, Wl = corr.shape[-2:]
        dx = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
        dy = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
        delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1)
        centroid = coords.reshape(B * H * W, 1, 1, 2) / 2 ** i
        coords_lvl = centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        c = _bilinear_sampler(corr.reshape(B * H * W, 1, Hl, Wl), coords_lvl)
        out.append(c.view(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def convex_upsample(flow, mask):
    """SegFlowGaussian.upsample_flow, SegFlowGaussian.py:846-857."""
    N, C, H, W = flow.shape
    mask = mask.view(N, 1, 9, 8, 8, H, W)
    mask = torch.softmax(mask, dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1)
    up = up.view(N, C, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2)
    up = up.permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, C, 8 * H, 8 * W)


# --------------------------------------------------------------------------- sliding window helpers
def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """SegmentationNetwork._compute_steps_for_sliding_window,
    nnunet/network_architecture/neural_network.py:267-290."""
    assert 0 < step_size <= 1
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step = image_size[dim] - patch_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def get_gaussian(patch_size, sigma_scale=1.0 / 8):
    """SegmentationNetwork._get_gaussian, neural_network.py:251-264."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(patch_size)
    center = [i // 2 for i in patch_size]
    sigmas = [i * sigma_scale for i in patch_size]
    tmp[tuple(center)] = 1
    g = gaussian_filter(tmp, sigmas, 0, mode="constant", cval=0)
    g = g / np.max(g) * 1
    g = g.astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_nd_image(image, new_shape=None, mode="constant", kwargs=None, return_slicer=False,
                 shape_must_be_divisible_by=None):
    """batchgenerators.augmentations.utils.pad_nd_image (>=0.23, un-vendored; PARITY UNPINNED).
    Call sites neural_network.py:309,442,476,644; SegFlowGaussian.py:3310-3313."""
    if kwargs is None:
        kwargs = {"constant_values": 0}
    old_shape = np.array(image.shape[-len(new_shape):]) if new_shape is not None else None
    if new_shape is None:
        assert shape_must_be_divisible_by is not None
        new_shape = image.shape[-len(shape_must_be_divisible_by):]
        old_shape = np.array(new_shape)
    num_axes_nopad = len(image.shape) - len(new_shape)
    new_shape = [max(new_shape[i], old_shape[i]) for i in range(len(new_shape))]
    if shape_must_be_divisible_by is not None:
        if not isinstance(shape_must_be_divisible_by, (list, tuple, np.ndarray)):
            shape_must_be_divisible_by = [shape_must_be_divisible_by] * len(new_shape)
        for i in range(len(new_shape)):
            if new_shape[i] % shape_must_be_divisible_by[i] != 0:
                new_shape[i] += shape_must_be_divisible_by[i] - new_shape[i] % shape_must_be_divisible_by[i]
    new_shape = np.array(new_shape)
    difference = new_shape - old_shape
    pad_below = difference // 2
    pad_above = difference // 2 + difference % 2
    pad_list = [[0, 0]] * num_axes_nopad + [list(i) for i in zip(pad_below, pad_above)]
    if not ((all(i == 0 for i in pad_below)) and (all(i == 0 for i in pad_above))):
        res = np.pad(image, pad_list, mode, **kwargs)
    else:
        res = image
    if not return_slicer:
        return res
    pad_arr = np.array(pad_list)
    pad_arr[:, 1] = np.array(res.shape) - pad_arr[:, 1]
    slicer = list(slice(*i) for i in pad_arr)
    return res, slicer


def normalize_intensity(x):
    """monai.transforms.NormalizeIntensity() defaults (nonzero=False, channel_wise=False): whole-array
    z-score with population std; only the mean is subtracted when std == 0.  monai is absent -> PARITY
    UNPINNED.  Call site SegFlowGaussian.py:3108."""
    m = x.mean()
    s = x.std(unbiased=False)
    return (x - m) / s if float(s) != 0.0 else x - m


# --------------------------------------------------------------------------- metrics of the parity bar
def dice(test, ref, label):
    """nnunet/evaluation/metrics.py:107-129: 2TP/(2TP+FP+FN); NaN when both empty."""
    t = np.asarray(test) == label
    r = np.asarray(ref) == label
    tp = float(np.sum(t & r))
    fp = float(np.sum(t & ~r))
    fn = float(np.sum(~t & r))
    if not (t.any() or r.any()):
        return float("nan")
    return 2.0 * tp / (2 * tp + fp + fn)


def mean_epe(a, b):
    """Mean end-point error between flows [...,2,H,W] (contract of SURVEY.md section 8d)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float(torch.sqrt(((a - b) ** 2).sum(dim=-3)).mean())


# --------------------------------------------------------------------------- export post-processing
def remove_all_but_the_largest_connected_component(image, for_which_classes, volume_per_voxel, minimum_valid_object_size=None):
    """nnunet/postprocessing/connected_components.py:51-107 (scipy.ndimage.label, default face connectivity)."""
    from scipy.ndimage import label
    if for_which_classes is None:
        for_which_classes = np.unique(image)
        for_which_classes = for_which_classes[for_which_classes > 0]
    assert 0 not in for_which_classes, "cannot remove background"
    largest_removed, kept_size = {}, {}
    for c in for_which_classes:
        if isinstance(c, (list, tuple)):
            c = tuple(c)
            mask = np.zeros_like(image, dtype=bool)
            for cl in c:
                mask[image == cl] = True
        else:
            mask = image == c
        lmap, num_objects = label(mask.astype(int))
        object_sizes = {i: (lmap == i).sum() * volume_per_voxel for i in range(1, num_objects + 1)}
        largest_removed[c] = None
        kept_size[c] = None
        if num_objects > 0:
            maximum_size = max(object_sizes.values())
            kept_size[c] = maximum_size
            for i in range(1, num_objects + 1):
                if object_sizes[i] != maximum_size:
                    remove = True
                    if minimum_valid_object_size is not None:
                        remove = object_sizes[i] < minimum_valid_object_size[c]
                    if remove:
                        image[(lmap == i) & mask] = 0
                        largest_removed[c] = object_sizes[i] if largest_removed[c] is None else max(largest_removed[c], object_sizes[i])
    return image, largest_removed, kept_size


def resample_data_or_seg(data, new_shape, is_seg, axis=None, order=3, do_separate_z=False, order_z=0):
    """nnunet/preprocessing/preprocessing.py:111-200 restated without skimage / batchgenerators (both absent: PARITY UNPINNED).
    skimage.transform.resize(order, mode='edge', anti_aliasing=False) and batchgenerators' resize_segmentation(order 0) both
    sample at src = scale*(dst+0.5)-0.5 with edge clamping = scipy map_coordinates(order, mode='nearest') on that map."""
    from scipy.ndimage import map_coordinates
    assert len(data.shape) == 4 and len(new_shape) == 3
    shape = np.array(data[0].shape)
    new_shape = np.array(new_shape)
    if not np.any(shape != new_shape):
        return data
    dtype_data = data.dtype
    orders = [order] * 3
    if do_separate_z:
        assert len(axis) == 1
        orders[int(axis[0])] = order_z
    out = []
    for c in range(data.shape[0]):
        vol = data[c].astype(float)
        # separable: resample one axis at a time (in-plane first, like the reference's per-slice resize followed by the z pass)
        for ax in sorted(range(3), key=lambda a: (do_separate_z and a == int(axis[0]) if axis is not None and len(axis) else False)):
            n_src, n_dst = vol.shape[ax], int(new_shape[ax])
            if n_src == n_dst:
                continue
            coords = (float(n_src) / n_dst) * (np.arange(n_dst) + 0.5) - 0.5
            grids = np.meshgrid(*[coords if a == ax else np.arange(vol.shape[a], dtype=float) for a in range(3)], indexing="ij")
            vol = map_coordinates(vol, np.array(grids), order=orders[ax], mode="nearest")
        out.append(vol[None].astype(dtype_data))
    return np.vstack(out)
