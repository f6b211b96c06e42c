"""Deterministic random-init weights keyed by the reference's ``state_dict`` names.

There are no checkpoints in the reference snapshot, so
tests and bench.py run every architecture with a seeded fill that depends only
on (parameter name, shape, seed).  The same fill loads into the product modules
(cineflow.*), the oracle modules (oracle.models) and -- in the build container --
the reference classes, because all three share the reference's key names
(SURVEY.md appendix A).
"""
import math
import zlib

import torch


def _kind_std(name, shape):
    """Return (kind, std) for one parameter; kinds: 'skip', 'matrix', 'scale', 'bias'."""
    if name.endswith("grid"):  # SpatialTransformer's persistent identity-grid buffer
        return "skip", 0.0
    if name.endswith(("num_batches_tracked", "relative_position_index", "attn_mask")):   # BatchNorm counter, Swin index / mask buffers
        return "skip", 0.0
    if name.endswith("running_var"):        # BatchNorm running variance: positive, away from zero
        return "var", 0.3
    if name.endswith("running_mean"):
        return "bias", 0.2
    if name.endswith("relative_position_bias_table"):
        return "matrix", 0.5
    if len(shape) >= 2:
        transposed = (".up.0." in name) or (".tu." in name) or name.startswith("tu.")
        fan_in = shape[0] * int(math.prod(shape[2:])) if transposed else int(math.prod(shape[1:]))
        if any(t in name for t in ("in_proj_weight", "out_proj", "linear1", "linear2")):
            return "matrix", math.sqrt(1.0 / fan_in)
        if "final_conv" in name:
            return "matrix", 0.05
        if "flow_head.conv2" in name:
            return "matrix", 0.1 * math.sqrt(2.0 / fan_in)
        return "matrix", math.sqrt(2.0 / fan_in)
    if name.endswith("weight"):
        return "scale", 0.1
    return "bias", 0.05


def seeded_tensor(name, shape, seed=0):
    kind, std = _kind_std(name, tuple(shape))
    if kind == "skip":
        return None
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if kind == "var":
        return 0.6 + std * t.abs()
    if kind == "scale":
        return 1.0 + std * t
    return std * t


def seeded_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    out = {}
    for name, shape in shapes.items():
        t = seeded_tensor(name, shape, seed)
        if t is not None:
            out[name] = t
    return out


def fill_module_(module, seed=0):
    """In-place seeded fill of a torch.nn.Module (oracle or reference instance)."""
    sd = module.state_dict()
    new = seeded_state_dict({k: v.shape for k, v in sd.items()}, seed)
    for k, v in new.items():
        sd[k] = v.to(sd[k].dtype)
    module.load_state_dict(sd, strict=True)
    return module
