"""Contour tracking and strain from the exported flow fields -- nnunet/get_strain.py on the HIP path.

get_strain.py reads `Postprocessed/Flow/<patient>/*.npz['flow']` ([H, W, D, 2] per frame), moves the ground-truth endo- / epicardial
/ RV contour points of a slice with the predicted ED->t flow and derives
  * the tracking error of the points against their ground-truth positions (`from_ed_accumulation` :51-140, `to_ed_accumulation`
    :141-232, `to_ed` :233-318), and
  * radial / circumferential Green-Lagrange strain curves and their smoothness (`from_ed` :319-433, `curvature` / `smoothness_measure`
    :26-38).
The one device operation is SpatialTransformerContour (network_architecture/integration.py:5-34): the flow field sampled bilinearly at
the contour points (cf_sample_points_2d); everything after it is arithmetic on a few hundred points, kept in torch on the host exactly as
the reference writes it.  File plumbing (.mat / nibabel / pickle readers, matplotlib) is not mirrored: the functions take arrays.
"""
import numpy as np
import torch

from . import ops


def curvature(x, y):
    """get_strain.py:26-32."""
    dx, dy = np.gradient(x), np.gradient(y)
    d2x, d2y = np.gradient(dx), np.gradient(dy)
    return np.abs((dx * d2y - dy * d2x) / ((dx ** 2 + dy ** 2) ** (3 / 2)))


def smoothness_measure(x, y):
    """get_strain.py:34-38: variance of the curvature of the curve (x, y)."""
    return np.var(curvature(x, y))


class SpatialTransformerContour:
    """network_architecture/integration.py:5-34.  forward(new_locs [B,2,1,P], original [B,C,H,W]) -> [B,C,1,P]: `original` sampled at
    the points; channel 0 of new_locs indexes the LAST axis of `original` (normalised by shape[~0] - 1), channel 1 the one before it.
    (The reference normalises new_locs in place; callers pass a clone, and so nothing is written back here.)"""

    def __init__(self, size, mode="bilinear"):
        self.shape, self.mode = tuple(size), mode

    def __call__(self, new_locs, original, mode="bilinear"):
        B, two, one, P = new_locs.shape
        assert two == 2 and one == 1 and tuple(original.shape[-2:]) == self.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        out = ops.sample_points(original.to(dev, dtype=torch.float32).contiguous(), new_locs.reshape(B, 2, P).to(dev, dtype=torch.float32).contiguous())
        return out.view(B, original.shape[1], 1, P).to(new_locs.device)

    forward = __call__


def prepare_flow(flow_frames, ed_number):
    """get_strain.py:61-70, :85-93: per-frame exported flows [H,W,D,2] -> [D,T,2,W,H] with a NaN frame inserted at the ED position and
    the frames rotated so that ED comes first.  Returns (flow, frame_indices)."""
    video = [np.asarray(f).transpose((2, 3, 0, 1)) for f in flow_frames]          # D, C, H, W
    flow = np.stack(video, axis=1).transpose(0, 1, 2, 4, 3)                        # D, T, C, W, H
    flow = np.insert(flow, ed_number, values=np.nan, axis=1)
    idx = np.arange(flow.shape[1])
    idx = np.concatenate([idx[idx >= ed_number], idx[idx < ed_number]])
    assert idx[0] == ed_number
    return flow[:, idx], idx


def track_from_ed(first_contours, slice_flow, spatial_transformer=None):
    """get_strain.py:383-397: first_contours [S,2,P] (structures x (coordinate) x points, ED frame), slice_flow [T,2,A,B] (frame 0 = ED,
    NaN) -> positions [T,S,2,P]: the ED points displaced by the ED->t flow sampled at the ED points."""
    st = spatial_transformer or SpatialTransformerContour(size=slice_flow.shape[-2:])
    first = first_contours[:, :, None, :]
    out = [torch.clone(first_contours)]
    for t in range(1, len(slice_flow)):
        delta = st(torch.clone(first), slice_flow[t][None].repeat(first.shape[0], 1, 1, 1))
        out.append((first + delta).squeeze(2))
    return torch.stack(out, dim=0)


def strain_curves(contour_points, to_roll=0):
    """get_strain.py:399-413: contour_points [T,2,2,P] in mm (endo, epi) -> (radial_strain [T], circ_strain [T]), Green-Lagrange
    0.5 (L^2 - L0^2) / L0^2 of the endo-epi distance and of the point-to-next-point distance along each contour (closed), averaged over
    the points, rolled back by `to_roll` frames to the acquisition order."""
    radial = torch.linalg.norm(torch.diff(contour_points, dim=1), dim=2).squeeze(1)                 # T, P
    radial_strain = 0.5 * ((radial ** 2 - radial[0][None] ** 2) / radial[0][None] ** 2)
    unfolded = torch.cat([contour_points, contour_points[:, :, :, 0][:, :, :, None]], dim=-1).unfold(-1, 2, 1)
    circ = torch.linalg.norm(torch.diff(unfolded, dim=-1), dim=2).squeeze(-1).mean(1)               # T, P
    circ_strain = 0.5 * ((circ ** 2 - circ[0][None] ** 2) / circ[0][None] ** 2)
    return (torch.roll(radial_strain, shifts=-to_roll, dims=[0]).mean(-1), torch.roll(circ_strain, shifts=-to_roll, dims=[0]).mean(-1))


def from_ed(slice_flow, contours, zoom, to_roll=0):
    """get_strain.py:319-433 for one slice: slice_flow [T,2,A,B] (ED first), contours [2,T,P,2] (endo / epi ground-truth points, already
    0-based and in the frame order of the flow), zoom = (mm, mm).  Returns dict(radial_strain, circ_strain, smooth)."""
    contours = torch.as_tensor(contours).float()
    first = contours[:, 0].permute(0, 2, 1)                                                          # 2, 2, P
    pts = track_from_ed(first, torch.as_tensor(slice_flow).float())                                  # T, 2, 2, P
    pts = pts * torch.as_tensor(np.asarray(zoom, dtype=np.float32).reshape(1, 1, 2, 1))            # mm
    radial, circ = strain_curves(pts, to_roll)
    smooth = (smoothness_measure(np.arange(len(radial)), radial.numpy()) + smoothness_measure(np.arange(len(circ)), circ.numpy())) / 2
    return {"radial_strain": radial, "circ_strain": circ, "smooth": float(smooth)}


def contour_tracking_error(slice_flow, contours, split_index, mode="from_ed_accumulation"):
    """The per-frame mean point error of get_strain.py's three tracking modes for one slice.

    slice_flow [T,2,A,B] (ED first, frame 0 NaN), contours [T,P,2] (0-based ground-truth points in the same frame order),
    split_index: np.cumsum([P_endo, P_epi]).  mode:
      'from_ed_accumulation' (:100-129): points of frame 0 pushed through the flows of frames 1..t one after the other;
      'to_ed_accumulation'   (:190-221): frames reversed, points of frame t pushed through flows t .. T-2;
      'to_ed'                (:283-306): frames reversed, one flow per frame.
    Returns [T-1, 3] (ENDO, EPI, RV)."""
    flow = torch.as_tensor(np.ascontiguousarray(slice_flow)).float()
    con = torch.as_tensor(np.ascontiguousarray(contours)).float()
    st = SpatialTransformerContour(size=flow.shape[-2:])
    errs = []
    if mode == "from_ed_accumulation":
        for t in range(1, len(flow)):
            cur = con[0].transpose(1, 0)[None, :, None, :]
            init = cur
            for t2 in range(1, t + 1):
                cur = cur + st(torch.clone(cur), flow[t2][None])
            delta = (cur - init).squeeze().permute(1, 0).numpy()
            errs.append(np.linalg.norm((con[t] - con[0]).numpy() - delta, axis=1))
        e = np.stack(errs, axis=0)
    else:
        flow, con = torch.flip(flow, dims=[0]), torch.flip(con, dims=[0])
        for t in range(len(flow) - 1):
            cur = con[t].transpose(1, 0)[None, :, None, :]
            init = cur
            steps = range(t, len(flow) - 1) if mode == "to_ed_accumulation" else [t]
            for t2 in steps:
                cur = cur + st(torch.clone(cur), flow[t2][None])
            delta = (cur - init).squeeze().permute(1, 0).numpy()
            errs.append(np.linalg.norm((con[-1] - con[t]).numpy() - delta, axis=1))
        e = np.flip(np.stack(errs, axis=0), axis=0)
    parts = np.split(e, indices_or_sections=split_index, axis=1)
    return np.stack([x.mean(-1) for x in parts], axis=-1)
