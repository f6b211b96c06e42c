"""CPU restatement of nnunet/get_strain.py's arithmetic and of SpatialTransformerContour (network_architecture/integration.py:5-34).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  SpatialTransformerContour and curvature / smoothness_measure are pinned against
the reference itself (tests/golden/make_golden_strain.py -> strain.npz); the strain curves and tracking errors restate :51-433."""
import numpy as np
import torch
import torch.nn.functional as F


def curvature(x, y):
    dx, dy = np.gradient(x), np.gradient(y)
    d2x, d2y = np.gradient(dx), np.gradient(dy)
    return np.abs((dx * d2y - dy * d2x) / ((dx ** 2 + dy ** 2) ** (3 / 2)))


def smoothness_measure(x, y):
    return np.var(curvature(x, y))


def spatial_transformer_contour(new_locs, original, shape):
    """integration.py:16-34 (new_locs is modified in place there; here on a copy)"""
    new_locs = new_locs.clone()
    for i in range(2):
        new_locs[:, i, ...] = 2 * (new_locs[:, i, ...] / (shape[~i] - 1) - 0.5)
    return F.grid_sample(original, new_locs.permute(0, 2, 3, 1), align_corners=True, mode="bilinear")


def track_from_ed(first_contours, slice_flow):
    first = first_contours[:, :, None, :]
    out = [torch.clone(first_contours)]
    for t in range(1, len(slice_flow)):
        delta = spatial_transformer_contour(first, slice_flow[t][None].repeat(first.shape[0], 1, 1, 1), slice_flow.shape[-2:])
        out.append((first + delta).squeeze(2))
    return torch.stack(out, dim=0)


def strain_curves(contour_points, to_roll=0):
    radial = torch.linalg.norm(torch.diff(contour_points, dim=1), dim=2).squeeze(1)
    radial_strain = 0.5 * ((radial ** 2 - radial[0][None] ** 2) / radial[0][None] ** 2)
    unfolded = torch.cat([contour_points, contour_points[:, :, :, 0][:, :, :, None]], dim=-1).unfold(-1, 2, 1)
    circ = torch.linalg.norm(torch.diff(unfolded, dim=-1), dim=2).squeeze(-1).mean(1)
    circ_strain = 0.5 * ((circ ** 2 - circ[0][None] ** 2) / circ[0][None] ** 2)
    return (torch.roll(radial_strain, shifts=-to_roll, dims=[0]).mean(-1), torch.roll(circ_strain, shifts=-to_roll, dims=[0]).mean(-1))


def from_ed(slice_flow, contours, zoom, to_roll=0):
    contours = torch.as_tensor(contours).float()
    first = contours[:, 0].permute(0, 2, 1)
    pts = track_from_ed(first, torch.as_tensor(slice_flow).float())
    pts = pts * torch.as_tensor(np.asarray(zoom, dtype=np.float32).reshape(1, 1, 2, 1))
    radial, circ = strain_curves(pts, to_roll)
    smooth = (smoothness_measure(np.arange(len(radial)), radial.numpy()) + smoothness_measure(np.arange(len(circ)), circ.numpy())) / 2
    return {"radial_strain": radial, "circ_strain": circ, "smooth": float(smooth)}


def contour_tracking_error(slice_flow, contours, split_index, mode="from_ed_accumulation"):
    flow = torch.as_tensor(np.ascontiguousarray(slice_flow)).float()
    con = torch.as_tensor(np.ascontiguousarray(contours)).float()
    shape = flow.shape[-2:]
    errs = []
    if mode == "from_ed_accumulation":
        for t in range(1, len(flow)):
            cur = con[0].transpose(1, 0)[None, :, None, :]
            init = cur
            for t2 in range(1, t + 1):
                cur = cur + spatial_transformer_contour(cur, flow[t2][None], shape)
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
                cur = cur + spatial_transformer_contour(cur, flow[t2][None], shape)
            delta = (cur - init).squeeze().permute(1, 0).numpy()
            errs.append(np.linalg.norm((con[-1] - con[t]).numpy() - delta, axis=1))
        e = np.flip(np.stack(errs, axis=0), axis=0)
    parts = np.split(e, indices_or_sections=split_index, axis=1)
    return np.stack([x.mean(-1) for x in parts], axis=-1)
