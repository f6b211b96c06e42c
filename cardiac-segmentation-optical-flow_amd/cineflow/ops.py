"""Thin tensor-level wrappers over the C ABI (include/cineflow.h).

PyTorch is used for device memory (torch.empty on the caching allocator) and the current HIP stream only; every
number is produced by a hand-written HIP kernel in libcineflow_hip.so.  Inputs must be CUDA(=HIP) tensors; nothing
here runs on the CPU and nothing falls back to torch operators.
"""
import os

import numpy as np
import torch

from ._lib import lib, check

ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5}
RES = {None: 0, "none": 0, "before_act": 1, "after_act": 2}
OP = {"add": 0, "sub": 1, "mul": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor (cineflow has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _u8(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
        raise TypeError("%s must be a contiguous CUDA uint8 tensor" % name)
    return t.data_ptr()


def _opt(t, name="tensor"):
    return None if t is None else _f32(t, name)


# ------------------------------------------------------------------------------------------------ warp family
def warp_bilinear(flow, src):
    """SpatialTransformer.forward: flow [B,2,H,W] / src [B,C,H,W], or the 3-D branch flow [B,3,D,H,W] / src [B,C,D,H,W]."""
    if src.dim() == 5:
        B, C, D, H, W = src.shape
        assert flow.shape == (B, 3, D, H, W), (flow.shape, src.shape)
        out = torch.empty_like(src)
        check(lib().cf_warp_trilinear_3d(_f32(flow), _f32(src), _f32(out), B, C, D, H, W, _stream()), "cf_warp_trilinear_3d")
        return out
    B, C, H, W = src.shape
    assert flow.shape == (B, 2, H, W), (flow.shape, src.shape)
    out = torch.empty_like(src)
    check(lib().cf_warp_bilinear_2d(_f32(flow), _f32(src), _f32(out), B, C, H, W, _stream()), "cf_warp_bilinear_2d")
    return out


def vecint(vec, nsteps=7):
    if vec.dim() == 5:   # 3-D fields: the same scaling and squaring, one warp launch per step
        assert vec.shape[1] == 3
        v = mul(vec, torch.full((1,), 1.0 / (2 ** nsteps), dtype=torch.float32, device=vec.device))
        for _ in range(nsteps):
            v = add(v, warp_bilinear(v, v))
        return v
    B, two, H, W = vec.shape
    assert two == 2
    out = torch.empty_like(vec)
    tmp = torch.empty_like(vec)
    check(lib().cf_vecint_2d(_f32(vec), _f32(out), _f32(tmp), B, H, W, nsteps, _stream()), "cf_vecint_2d")
    return out


def warp_labels(flow, labels, num_classes=4):
    """flow [T,B,2,H,W] float32; labels uint8 [B,H,W] -> uint8 [T,B,H,W]."""
    T, B, two, H, W = flow.shape
    assert two == 2 and labels.shape == (B, H, W)
    out = torch.empty((T, B, H, W), dtype=torch.uint8, device=flow.device)
    check(lib().cf_warp_labels_2d(_f32(flow), _u8(labels), _u8(out), T, B, num_classes, H, W, _stream()), "cf_warp_labels_2d")
    return out


def memory_input(x0, xt, cum):
    B, one, H, W = x0.shape
    assert one == 1 and xt.shape == x0.shape and cum.shape == (B, 2, H, W)
    out = torch.empty((B, 6, H, W), dtype=torch.float32, device=x0.device)
    check(lib().cf_memory_input(_f32(x0), _f32(xt), _f32(cum), _f32(out), B, H, W, _stream()), "cf_memory_input")
    return out


def jacobian_det(disp):
    """disp [B,2,H,W] float32 -> float64 [B,H,W]; or the 3-D case disp [B,3,D,H,W] -> float64 [B,D,H,W]."""
    if disp.dim() == 5:
        B, three, D, H, W = disp.shape
        assert three == 3
        out = torch.empty((B, D, H, W), dtype=torch.float64, device=disp.device)
        check(lib().cf_jacobian_det_3d(_f32(disp), out.data_ptr(), B, D, H, W, _stream()), "cf_jacobian_det_3d")
        return out
    B, two, H, W = disp.shape
    assert two == 2
    out = torch.empty((B, H, W), dtype=torch.float64, device=disp.device)
    check(lib().cf_jacobian_det_2d(_f32(disp), out.data_ptr(), B, H, W, _stream()), "cf_jacobian_det_2d")
    return out


# ------------------------------------------------------------------------------------------------ correlation
def corr_volume(cur, prev, radius=4, stride=1):
    B, C, H, W = cur.shape
    assert prev.shape == cur.shape
    out = torch.empty((B, (2 * radius + 1) ** 2, H, W), dtype=torch.float32, device=cur.device)
    check(lib().cf_corr_volume(_f32(cur), _f32(prev), _f32(out), B, C, H, W, radius, stride, _stream()), "cf_corr_volume")
    return out


def pyramid_numel(B, H, W, levels):
    N = H * W
    return sum(B * N * (H >> l) * (W >> l) for l in range(levels))


def corr_pyramid(f1, f2, levels=4):
    B, C, H, W = f1.shape
    assert f2.shape == f1.shape
    pyr = torch.empty(pyramid_numel(B, H, W, levels), dtype=torch.float32, device=f1.device)
    check(lib().cf_corr_pyramid(_f32(f1), _f32(f2), _f32(pyr), B, C, H, W, levels, _stream()), "cf_corr_pyramid")
    return pyr


def corr_lookup(pyr, coords, levels=4, radius=4):
    B, two, H, W = coords.shape
    assert two == 2 and pyr.numel() == pyramid_numel(B, H, W, levels)
    out = torch.empty((B, levels * (2 * radius + 1) ** 2, H, W), dtype=torch.float32, device=coords.device)
    check(lib().cf_corr_lookup(_f32(pyr), _f32(coords), _f32(out), B, H, W, levels, radius, _stream()), "cf_corr_lookup")
    return out


def convex_upsample(flow, mask):
    B, C, h, w = flow.shape
    assert mask.shape == (B, 576, h, w)
    out = torch.empty((B, C, 8 * h, 8 * w), dtype=torch.float32, device=flow.device)
    check(lib().cf_convex_upsample(_f32(flow), _f32(mask), _f32(out), B, C, h, w, _stream()), "cf_convex_upsample")
    return out


# ------------------------------------------------------------------------------------------------ conv / norm / attention
def prep_conv_weight(w):
    """torch conv weight [Cout,Cin,KH,KW] -> Wt [Cin*KH*KW, Cout] (done once at model load)."""
    cout = w.shape[0]
    return w.reshape(cout, -1).t().contiguous()


# ---- f16 hi/lo-split convolution (conv_f16s.hip) ------------------------------------------------------------------
CONV_MODE = "f16s"   # "f16s": f16-MFMA 3-term split where supported, exact fp32 MFMA elsewhere;  "f32": fp32 MFMA only


def set_conv_mode(mode):
    global CONV_MODE
    assert mode in ("f16s", "f32")
    CONV_MODE = mode


import contextlib


@contextlib.contextmanager
def conv_terms(terms):
    """Product mode of the f16-MFMA convolutions launched from this thread inside the block (cf_conv_terms): 3 = hi/lo split (f32-class,
    the default), 1 = hi x hi only -- operands rounded to fp16, fp32 accumulation: the reference's fp16 autocast on the segmentation path
    (mixed_precision=True, neural_network.py:140-146).  The flow path never runs in mode 1 (SegFlowGaussian.py:2905-2909 forces it off)."""
    prev = lib().cf_conv_terms(int(terms))
    try:
        yield
    finally:
        lib().cf_conv_terms(prev)


def f16s_supported(kh, kw, stride, pad):
    """kernel shapes of conv_f16s.hip: 3x3 pad 1 and 1x1 pad 0 at stride 1 / 2; the separable 1x5 pad (0,2) / 5x1 pad (2,0) of RAFT's
    SepConvGRU at stride 1.  (The 7x7 convolution of the 2-channel flow in RAFT's motion encoder stays on the exact fp32 kernel by
    design: padding 2 channels to a 16-channel chunk would do 8x the work.)"""
    pad = tuple(pad)
    if stride in (1, 2) and ((kh, kw, pad) == (3, 3, (1, 1)) or (kh, kw, pad) == (1, 1, (0, 0))):
        return True
    return stride == 1 and ((kh, kw, pad) == (1, 5, (0, 2)) or (kh, kw, pad) == (5, 1, (2, 0)))


def f16s_chunk(kh, kw):
    """channels per LDS chunk of the kernel shape (16 for 3x3, 32 otherwise)"""
    return 16 if (kh, kw) == (3, 3) else 32


def f16s_dynamic_ok(x1, x2, kh, kw=None, out_sample_elems=None, out_hw=None):
    """Run-time limits of conv_f16s.hip (conv_f16s_supported): one sample of each input below 2 GiB (32-bit buffer offsets; larger BATCHES
    are split inside the library) and -- when the caller passes them -- one output sample (all `out_ctotal` channels of the destination
    buffer, x4 for the transposed convolution's scatter) below 1 GiB, 8 of them when an output image has <= 256 pixels (the epilogue's
    32-bit store offsets).  A cat[x1, x2] whose split is not a chunk multiple is handled by split-aware weight packing
    (pack_conv_weight_f16s(w, c1=...)), not by another kernel."""
    per = lambda t: 0 if t is None else (t.numel() // t.shape[0]) * 4
    if out_sample_elems is not None and out_sample_elems * 4 * (8 if (out_hw is not None and out_hw <= 256) else 1) >= 2 ** 30:
        return False
    return per(x1) < 2 ** 31 and per(x2) < 2 ** 31


def pack_conv_weight_f16s(w, c1=None):
    """torch conv weight [Cout,Cin,KH,KW] (3x3, 1x1, 1x5, 5x1) -> (packed fp16 tensor, scale exponent s).

    Fragment order [m-tile][chunk][tap][kstep][part hi/lo][lane = h*32 + r][j]: value = 2^s * W[mt*32 + r][chunk*CK + kstep*16 + 8h + j][tap],
    CK = 16 (3x3) or 32; m-tiles padded to an even count when Cout > 32, channels padded to CK; the power of two
    2^s brings max|W| to ~2^10 so that the lo halves stay in fp16's normal range (exact scaling, undone through alpha).
    c1: the input is cat[x1 (c1 channels), x2]: x1's channels are padded to whole chunks (zero weights), x2's follow -- the kernel
    switches input pointers at a chunk boundary."""
    import math
    cout, cin, kh, kw = w.shape
    ntap = kh * kw
    ck = f16s_chunk(kh, kw)
    ks = ck // 16
    nmt = 1 if cout <= 32 else 2 * ((cout + 63) // 64)
    wmax = float(w.abs().max())
    s = int(math.floor(math.log2(1024.0 / wmax))) if wmax > 0 else 0
    s = max(-24, min(24, s))
    ws = w.reshape(cout, cin, ntap).to(torch.float32) * (2.0 ** s)
    if c1 is not None and 0 < c1 < cin and c1 % ck:
        c1p = (c1 + ck - 1) // ck * ck
        nchunk = c1p // ck + (cin - c1 + ck - 1) // ck
        wp = torch.zeros((nmt * 32, nchunk * ck, ntap), dtype=torch.float32, device=w.device)
        wp[:cout, :c1] = ws[:, :c1]
        wp[:cout, c1p:c1p + cin - c1] = ws[:, c1:]
    else:
        nchunk = (cin + ck - 1) // ck
        wp = torch.zeros((nmt * 32, nchunk * ck, ntap), dtype=torch.float32, device=w.device)
        wp[:cout, :cin] = ws
    hi = wp.half()
    lo = (wp - hi.float()).half()
    x = torch.stack([hi, lo])                                  # [part, co, ci, tap]
    x = x.view(2, nmt, 32, nchunk, ks, 2, 8, ntap)             # part, mt, r, chunk, ks, h, j, tap
    x = x.permute(1, 3, 7, 4, 0, 5, 2, 6).contiguous()        # mt, chunk, tap, ks, part, h, r, j
    return x.view(-1), s


# ---- row Winograd F(2,3) form of the same convolution (conv_wino.hip) ----------------------------------------------------
WINO = os.environ.get("CF_CONV_WINO", "1") != "0"         # 0: every layer stays on the direct kernels (A/B knob; the library reads it too)
_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))      # G of F(2,3): U = G g along kx


def wino_ok(B, C1, C2, H, W, cout, prenorm=False):
    """does cf_conv2d_wino take this 3x3 / stride 1 / pad 1 layer?  (Cout in whole 128-channel blocks or a last block >= 96, W % 16 == 0,
    H a multiple of the tile rows)"""
    return bool(WINO and CONV_MODE == "f16s" and lib().cf_conv2d_wino_ok(B, C1, C2, H, W, cout, 1 if prenorm else 0) == 1)


def pack_conv_weight_wino(w, c1=None):
    """torch conv weight [Cout,Cin,3,3] -> (packed fp16 tensor, scale exponent s) for cf_conv2d_wino.

    U[co][ci][ky][pos] = sum_kx G[pos][kx] w[co][ci][ky][kx] in fp64 (G = the F(2,3) kernel transform), scaled by 2^s (max|U| -> ~2^10),
    rounded to fp32 and split into fp16 hi / lo.  Fragment order [m-tile][chunk][step = ky * 4 + pos][part hi/lo][lane = h*32 + r][j]:
    value = 2^s * U[mt*32 + r][chunk*16 + 8h + j][ky][pos]; m-tiles padded to whole 128-channel blocks, channels to 16; c1 as in
    pack_conv_weight_f16s (cat[x1, x2] with x1's channels padded to whole chunks)."""
    import math
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    U = torch.einsum("pk,ocyk->ocyp", G, w.to(torch.float64)).reshape(cout, cin, 12)
    umax = float(U.abs().max())
    s = int(math.floor(math.log2(1024.0 / umax))) if umax > 0 else 0
    s = max(-24, min(24, s))
    us = (U * (2.0 ** s)).to(torch.float32)
    ck = 16
    nmt = 4 * ((cout + 127) // 128)
    if c1 is not None and 0 < c1 < cin and c1 % ck:
        c1p = (c1 + ck - 1) // ck * ck
        nchunk = c1p // ck + (cin - c1 + ck - 1) // ck
        wp = torch.zeros((nmt * 32, nchunk * ck, 12), dtype=torch.float32, device=w.device)
        wp[:cout, :c1] = us[:, :c1]
        wp[:cout, c1p:c1p + cin - c1] = us[:, c1:]
    else:
        nchunk = (cin + ck - 1) // ck
        wp = torch.zeros((nmt * 32, nchunk * ck, 12), dtype=torch.float32, device=w.device)
        wp[:cout, :cin] = us
    hi = wp.half()
    lo = (wp - hi.float()).half()
    x = torch.stack([hi, lo])                                  # [part, co, ci, step]
    x = x.view(2, nmt, 32, nchunk, 2, 8, 12)                   # part, mt, r, chunk, h, j, step
    x = x.permute(1, 3, 6, 0, 4, 2, 5).contiguous()           # mt, chunk, step, part, h, r, j
    return x.view(-1), s


def conv2d_wino(x1, wpk, wscale, bias, cout, x2=None, act=None, res=None, out=None, out_coff=0, alpha=1.0, stats_groups=None):
    """cf_conv2d_wino: 3x3 / stride 1 / pad 1; arguments and returns as conv2d_f16s (wpk from pack_conv_weight_wino)."""
    B, C1, H, W = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    if out is None:
        out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x1.device)
    assert out.shape[0] == B and out.shape[2] == H and out.shape[3] == W
    if res is not None:
        assert res.shape == (B, cout, H, W)
    assert wpk.dtype == torch.float16 and wpk.is_cuda
    if F16S_RANGE_CHECK:
        _range_check(x1, x2)
    ws = _zeroed_stats_ws(2 * B * stats_groups, x1.device) if stats_groups else None
    check(lib().cf_conv2d_wino(_f32(x1), C1, _opt(x2), C2, wpk.data_ptr(), _opt(bias), _opt(res), _f32(out), out.shape[1], out_coff, B, H, W,
                               cout, ACT[act], float(alpha) * (2.0 ** -wscale), None if ws is None else ws.data_ptr(),
                               -stats_groups if stats_groups else 0, _stream()), "cf_conv2d_wino")
    return (out, ws) if stats_groups else out


def conv2d_wino_prenorm(x, coef, slope, wpk, wscale, bias, cout, stats_groups=None):
    """cf_conv2d_wino_prenorm: as conv2d_f16s_prenorm on the Winograd kernel."""
    B, C, H, W = x.shape
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    ws = _zeroed_stats_ws(2 * B * stats_groups, x.device) if stats_groups else None
    check(lib().cf_conv2d_wino_prenorm(_f32(x), C, _f32(coef), float(slope), wpk.data_ptr(), _opt(bias), _f32(out), B, H, W, cout, 2.0 ** -wscale,
                                       None if ws is None else ws.data_ptr(), -stats_groups if stats_groups else 0, _stream()), "cf_conv2d_wino_prenorm")
    return (out, ws) if stats_groups else out


# CF_F16S_RANGE_CHECK=1 (debug): every input of an f16-split convolution is scanned for values the hi/lo split cannot carry (non-finite or
# |x| >= 65504, e.g. a `noNorm` modality fed to a network without an input normalisation); f16s_range_violations() returns the count so far.
# Such values come out of the kernel as NaN by design; the exact route for them is set_conv_mode("f32").
F16S_RANGE_CHECK = os.environ.get("CF_F16S_RANGE_CHECK", "0") == "1"
_range_counter = {}


def _range_check(*tensors):
    for t in tensors:
        if t is None:
            continue
        c = _range_counter.get(t.device)
        if c is None:
            c = _range_counter[t.device] = torch.zeros(1, dtype=torch.int64, device=t.device)
        check(lib().cf_count_out_of_range(_f32(t), t.numel(), 65504.0, c.data_ptr(), _stream()), "cf_count_out_of_range")


def f16s_range_violations(reset=False):
    """number of f16-split convolution input elements seen outside the supported range since the last reset (0 unless F16S_RANGE_CHECK)"""
    n = sum(int(c.item()) for c in _range_counter.values())
    if reset:
        for c in _range_counter.values():
            c.zero_()
    return n


_zero_pool = {}


def _zeroed_stats_ws(n, device):
    """n doubles of a pool cleared with ONE memset (the convolution's fused GroupNorm statistics accumulate with atomics and need a
    zero start; a memset per launch was 1.2 % of the step).  Slices are handed out once; an exhausted pool is replaced by a fresh
    zeroed one (the old one lives as long as its slices do)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    pool = _zero_pool.get(key)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = [torch.zeros(max(n, 1 << 23), dtype=torch.float64, device=device), 0]   # 64 MB
        _zero_pool[key] = pool
    ws = pool[0][pool[1]:pool[1] + n]
    pool[1] += (n + 1) & ~1   # keep 16-byte alignment
    return ws


def conv2d_f16s(x1, wpk, wscale, bias, cout, kh, kw, stride=1, pad=(0, 0), x2=None, act=None, res=None, out=None, out_coff=0, alpha=1.0,
                stats_groups=None):
    """With stats_groups=G the call returns (out, ws): ws holds the GroupNorm statistics of `out` for group_norm_apply."""
    B, C1, H, W = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    Ho = (H + 2 * pad[0] - kh) // stride + 1
    Wo = (W + 2 * pad[1] - kw) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x1.device)
    assert out.shape[0] == B and out.shape[2] == Ho and out.shape[3] == Wo
    if res is not None:
        assert res.shape == (B, cout, Ho, Wo)
    assert wpk.dtype == torch.float16 and wpk.is_cuda
    if F16S_RANGE_CHECK:
        _range_check(x1, x2)
    ws = _zeroed_stats_ws(2 * B * stats_groups, x1.device) if stats_groups else None
    check(lib().cf_conv2d_f16s(_f32(x1), C1, _opt(x2), C2, wpk.data_ptr(), _opt(bias), _opt(res), _f32(out), out.shape[1], out_coff, B, H, W,
                               cout, kh, kw, stride, pad[0], pad[1], ACT[act], float(alpha) * (2.0 ** -wscale),
                               None if ws is None else ws.data_ptr(), -stats_groups if stats_groups else 0, _stream()), "cf_conv2d_f16s")
    return (out, ws) if stats_groups else out


DIRECT_STEM = os.environ.get("CF_CONV_DIRECT", "1") != "0"      # 0: the stems go through the MFMA kernels like every other layer (A/B knob)


def small_cin_supported(cin, kh, kw, stride, pad, stats_groups=None):
    """layers routed to cf_conv2d_small_cin: 1 or 2 input channels (3x3 pad 1 or 1x1) and 6 input channels 1x1, stride 1, <= 64
    statistics groups.  Measured at 256x256 (tools/microbench.py --only stem): 1 -> 32 B120 209 us vs 641 us on the MFMA kernel,
    1 -> 64 106 vs 238, 6 -> 64 1x1 120 vs 194; the 6 -> 64 3x3 layer is FMA-bound in the direct kernel (304 vs 257 us) and stays
    on the MFMA kernel."""
    if not DIRECT_STEM or stride != 1 or (stats_groups and stats_groups > 64):
        return False
    k3, k1 = (kh, kw, tuple(pad)) == (3, 3, (1, 1)), (kh, kw, tuple(pad)) == (1, 1, (0, 0))
    return (cin in (1, 2) and (k3 or k1)) or (cin == 6 and k1)


def conv2d_small_cin(x, weight, bias, stats_groups=None):
    """Stem convolution as a direct fp32 kernel: x [B,Cin,H,W], weight [Cout,Cin,K,K] (checkpoint layout).  With stats_groups=G the
    call returns (out, ws) like conv2d_f16s."""
    B, Cin, H, W = x.shape
    cout, cin_w, K, K2 = weight.shape
    assert cin_w == Cin and K == K2
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * B * stats_groups, dtype=torch.float64, device=x.device) if stats_groups else None
    check(lib().cf_conv2d_small_cin(_f32(x), _f32(weight), _opt(bias), _f32(out), B, Cin, H, W, cout, K, None if ws is None else ws.data_ptr(),
                                    stats_groups or 0, _stream()), "cf_conv2d_small_cin")
    return (out, ws) if stats_groups else out


def small_cout_supported(cout, kh, kw, stride, pad):
    """layers routed to cf_conv2d_small_cout: 3x3 / pad 1 / stride 1 with at most 4 output channels (the flow heads).  Measured at
    128 x 64 x 256 x 256 -> 2: see DESIGN.md (the MFMA kernel fills 2 of its 32 rows: 16 TF)."""
    return DIRECT_STEM and cout <= 4 and (kh, kw, tuple(pad)) == (3, 3, (1, 1)) and stride == 1


def conv2d_small_cout(x, weight, bias, res=None):
    """Direct exact-fp32 3x3 convolution to <= 4 channels: x [B,Cin,H,W], weight [Cout,Cin,3,3] (checkpoint layout) -> conv + bias (+ res)."""
    B, Cin, H, W = x.shape
    cout = weight.shape[0]
    assert tuple(weight.shape) == (cout, Cin, 3, 3)
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.shape == out.shape
    check(lib().cf_conv2d_small_cout(_f32(x), _f32(weight), _opt(bias), _opt(res), _f32(out), B, Cin, H, W, cout, _stream()), "cf_conv2d_small_cout")
    return out


PRENORM = os.environ.get("CF_PRENORM", "1") != "0"        # 0: every normalisation runs as its own apply pass (A/B knob)


def prenorm_ok(x, cout):
    """can cf_conv2d_f16s_prenorm take this input (raw conv output [B,C,H,W]) for a 3x3 / stride 1 convolution to `cout` channels?"""
    B, C, H, W = x.shape
    return bool(PRENORM and CONV_MODE == "f16s" and x.data_ptr() % 16 == 0 and lib().cf_conv2d_f16s_prenorm_ok(B, C, H, W, cout) == 1)


def group_norm_coef(ws, gamma, beta, groups, B, C, HW, eps=1e-5):
    """{mean, rstd * gamma, beta} per (sample, channel) from a statistics workspace -> float [B,3,C] for conv2d_f16s_prenorm"""
    coef = torch.empty((B, 3, C), dtype=torch.float32, device=ws.device)
    check(lib().cf_group_norm_coef(ws.data_ptr(), _opt(gamma), _opt(beta), B, C, HW, groups, float(eps), _f32(coef), _stream()), "cf_group_norm_coef")
    return coef


def norm_head_ok(x, K):
    """can cf_norm_head_1x1 take the raw map x [B,C,H,W] into K classes?"""
    return x.dim() == 4 and K in (2, 4, 8) and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[1] <= 1024 and x.is_contiguous()


def norm_head_1x1(x, coef, slope, w, bias=None):
    """out[b,k] = bias[k] + sum_c w[k,c] * lrelu((x[b,c] - mean) * scale + shift, slope): the deferred InstanceNorm + LeakyReLU of a raw
    convolution output x [B,C,H,W] (coef from group_norm_coef) folded into a 1x1 head w [K,C] (Generic_UNet's seg_outputs[-1])."""
    B, C, H, W = x.shape
    K = w.shape[0]
    w2 = w.reshape(K, C).contiguous()
    out = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
    check(lib().cf_norm_head_1x1(_f32(x), _f32(coef), float(slope), _f32(w2), _opt(bias), _f32(out), B, C, H * W, K, _stream()), "cf_norm_head_1x1")
    return out


def conv2d_f16s_prenorm(x, coef, slope, wpk, wscale, bias, cout, stats_groups=None):
    """3x3 / stride 1 convolution of lrelu((x - mean) * scale + shift, slope): x is the producer's raw output, coef its
    group_norm_coef.  Returns (out, ws) with stats_groups like conv2d_f16s."""
    B, C, H, W = x.shape
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    ws = _zeroed_stats_ws(2 * B * stats_groups, x.device) if stats_groups else None
    check(lib().cf_conv2d_f16s_prenorm(_f32(x), C, _f32(coef), float(slope), wpk.data_ptr(), _opt(bias), _f32(out), B, H, W, cout, 2.0 ** -wscale,
                                       None if ws is None else ws.data_ptr(), -stats_groups if stats_groups else 0, _stream()), "cf_conv2d_f16s_prenorm")
    return (out, ws) if stats_groups else out


def conv_transpose2d_k2s2_f16s(x, wpk, wscale, bias, cout, out=None, out_coff=0, stats_groups=None):
    B, Cin, H, W = x.shape
    if F16S_RANGE_CHECK:
        _range_check(x)
    if out is None:
        out = torch.empty((B, cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    ws = _zeroed_stats_ws(2 * B * stats_groups, x.device) if stats_groups else None       # fused into the scatter epilogue (atomics: zero start)
    check(lib().cf_conv_transpose2d_k2s2_f16s(_f32(x), wpk.data_ptr(), _opt(bias), _f32(out), out.shape[1], out_coff, B, Cin, H, W, cout,
                                              2.0 ** -wscale, None if ws is None else ws.data_ptr(), -stats_groups if stats_groups else 0, _stream()),
          "cf_conv_transpose2d_k2s2_f16s")
    return (out, ws) if stats_groups else out


def conv2d(x1, wt, bias, cout, kh, kw, stride=1, pad=(0, 0), x2=None, act=None, res=None, out=None, out_coff=0, alpha=1.0,
           w_bstride=0):
    """act(alpha*conv(cat[x1,x2]) + bias) + res, written into channels [out_coff, out_coff+cout) of `out`."""
    B, C1, H, W = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None:
        assert x2.shape[0] == B and x2.shape[2:] == x1.shape[2:]
    assert wt.shape == ((C1 + C2) * kh * kw, cout) or w_bstride, (wt.shape, C1, C2, kh, kw, cout)
    Ho = (H + 2 * pad[0] - kh) // stride + 1
    Wo = (W + 2 * pad[1] - kw) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x1.device)
    assert out.shape[0] == B and out.shape[2] == Ho and out.shape[3] == Wo
    if res is not None:
        assert res.shape == (B, cout, Ho, Wo)
    check(lib().cf_conv2d(_f32(x1), C1, _opt(x2), C2, _f32(wt), w_bstride, _opt(bias), _opt(res), _f32(out), out.shape[1], out_coff,
                          B, H, W, cout, kh, kw, stride, pad[0], pad[1], ACT[act], float(alpha), _stream()), "cf_conv2d")
    return out


def conv_transpose2d_k2s2(x, w, bias, out=None, out_coff=0):
    """w in torch layout [Cin,Cout,2,2]."""
    B, Cin, H, W = x.shape
    assert w.shape[0] == Cin and w.shape[2:] == (2, 2)
    cout = w.shape[1]
    if out is None:
        out = torch.empty((B, cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    check(lib().cf_conv_transpose2d_k2s2(_f32(x), _f32(w), _opt(bias), _f32(out), out.shape[1], out_coff, B, Cin, H, W, cout,
                                         _stream()), "cf_conv_transpose2d_k2s2")
    return out


_ws_cache = {}


def _stats_ws(n, device):
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 4096), dtype=torch.float64, device=device)
        _ws_cache[key] = ws
    return ws


def group_norm(x, gamma, beta, groups, eps=1e-5, act=None, res=None, res_mode=None, out=None):
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    if out is None:
        out = torch.empty_like(x)
    ws = _stats_ws(2 * B * groups, x.device)
    check(lib().cf_group_norm(_f32(x), _opt(gamma), _opt(beta), _opt(res), _f32(out), B, C, HW, groups, float(eps), ACT[act],
                              RES[res_mode if res is not None else None], ws.data_ptr(), _stream()), "cf_group_norm")
    return out


def group_norm_apply(x, gamma, beta, groups, ws, eps=1e-5, act=None, res=None, res_mode=None, out=None, res_norm=None):
    """Apply pass only; `ws` from conv2d_f16s(..., stats_groups=groups).  res_norm=(ws_r, gamma_r, beta_r): `res` is a raw
    convolution output whose own GroupNorm (same groups / eps) is applied on the fly while it is added."""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    if out is None:
        out = torch.empty_like(x)
    assert ws.dtype == torch.float64 and ws.numel() >= 2 * B * groups
    if res_norm is not None:
        ws_r, gamma_r, beta_r = res_norm
        assert res is not None and res_mode in ("before_act", "after_act") and res.shape == x.shape
        assert ws_r.dtype == torch.float64 and ws_r.numel() >= 2 * B * groups
        check(lib().cf_group_norm_apply_res_norm(_f32(x), _opt(gamma), _opt(beta), _f32(res), _f32(out), B, C, HW, groups, float(eps), ACT[act],
                                                 RES[res_mode], ws.data_ptr(), ws_r.data_ptr(), _opt(gamma_r), _opt(beta_r), _stream()),
              "cf_group_norm_apply_res_norm")
        return out
    check(lib().cf_group_norm_apply(_f32(x), _opt(gamma), _opt(beta), _opt(res), _f32(out), B, C, HW, groups, float(eps), ACT[act],
                                    RES[res_mode if res is not None else None], ws.data_ptr(), _stream()), "cf_group_norm_apply")
    return out


def layer_norm_cf(x, gamma, beta, eps=1e-5, out=None):
    B, C, N = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().cf_layer_norm_cf(_f32(x), _f32(gamma), _f32(beta), _f32(out), B, C, N, float(eps), _stream()), "cf_layer_norm_cf")
    return out


def _cf_slice(t, name):
    """[B,C,N] tensor that is either contiguous or a channel slice (narrow on dim 1) of a contiguous buffer."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
        raise TypeError("%s must be a 3-D CUDA float32 tensor" % name)
    B, C, N = t.shape
    if t.stride(2) != 1 or t.stride(1) != N:
        raise ValueError("%s must be contiguous in its last two dims" % name)
    return t.data_ptr(), (t.stride(0) if B > 1 else max(t.stride(0), C * N))


def attention_cf(q, k, v, heads):
    """q [B,C,Nq], k/v [B,C,Nk] channel-first; each may be a channel slice of a fused projection buffer.  Token counts that are not
    multiples of 32 (e.g. the 28 x 28 bottleneck of a 224 x 224 image) are zero-padded here and the padded keys masked in the kernel."""
    B, C, Nq = q.shape
    Nk = k.shape[2]
    assert k.shape == (B, C, Nk) and v.shape == (B, C, Nk) and C % heads == 0
    if Nq % 32 or Nk % 32:
        Nqp, Nkp = (Nq + 31) // 32 * 32, (Nk + 31) // 32 * 32

        def padded(t, n):
            p = torch.zeros((B, C, n), dtype=torch.float32, device=t.device)
            p[:, :, :t.shape[2]] = t
            return p
        qp_, kp_, vp_ = padded(q, Nqp), padded(k, Nkp), padded(v, Nkp)
        out = torch.empty((B, C, Nqp), dtype=torch.float32, device=q.device)
        check(lib().cf_attention_cf_masked(_f32(qp_), C * Nqp, _f32(kp_), C * Nkp, _f32(vp_), C * Nkp, _f32(out), B, heads, C // heads, Nqp, Nkp, Nk,
                                           _stream()), "cf_attention_cf_masked")
        return out[:, :, :Nq].contiguous()
    qp, qbs = _cf_slice(q, "q")
    kp, kbs = _cf_slice(k, "k")
    vp, vbs = _cf_slice(v, "v")
    out = torch.empty((B, C, Nq), dtype=torch.float32, device=q.device)
    check(lib().cf_attention_cf(qp, qbs, kp, kbs, vp, vbs, _f32(out), B, heads, C // heads, Nq, Nk, _stream()), "cf_attention_cf")
    return out


# ------------------------------------------------------------------------------------------------ plumbing kernels
def gru_reset_mul(gates, h):
    B, C = h.shape[0], h.shape[1]
    HW = h.numel() // (B * C)
    out = torch.empty_like(h)
    check(lib().cf_gru_reset_mul(_f32(gates), _f32(h), _f32(out), B, C, HW, _stream()), "cf_gru_reset_mul")
    return out


def gru_blend(gates, h, cand):
    B, C = h.shape[0], h.shape[1]
    HW = h.numel() // (B * C)
    out = torch.empty_like(h)
    check(lib().cf_gru_blend(_f32(gates), _f32(h), _f32(cand), _f32(out), B, C, HW, _stream()), "cf_gru_blend")
    return out


def binary(op, a, b, out=None):
    """a op b with b broadcast over the leading dims of a (b.numel() divides a.numel())."""
    n, period = a.numel(), b.numel()
    assert n % period == 0
    if out is None:
        out = torch.empty_like(a)
    check(lib().cf_binary(OP[op], _f32(a), _f32(b), _f32(out), n, period, _stream()), "cf_binary")
    return out


def add(a, b, out=None):
    return binary("add", a, b, out)


def sub(a, b, out=None):
    return binary("sub", a, b, out)


def mul(a, b, out=None):
    return binary("mul", a, b, out)


def copy_channels(src, src_coff, C, dst=None, dst_coff=0, act=None):
    B, sct = src.shape[0], src.shape[1]
    HW = src.numel() // (B * sct)
    if dst is None:
        dst = torch.empty((B, C) + tuple(src.shape[2:]), dtype=torch.float32, device=src.device)
    check(lib().cf_copy_channels(_f32(src), sct, src_coff, _f32(dst), dst.shape[1], dst_coff, B, C, HW, ACT[act], _stream()),
          "cf_copy_channels")
    return dst


def coords_grid(B, H, W, device):
    out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
    check(lib().cf_coords_grid(_f32(out), B, H, W, _stream()), "cf_coords_grid")
    return out


def crop2d(src, y0, x0, h, w):
    lead, (H, W) = src.shape[:-2], src.shape[-2:]
    N = src.numel() // (H * W)
    dst = torch.empty(tuple(lead) + (h, w), dtype=torch.float32, device=src.device)
    check(lib().cf_crop2d(_f32(src), _f32(dst), N, H, W, y0, x0, h, w, _stream()), "cf_crop2d")
    return dst


def pad2d(src, y0, x0, H, W):
    lead, (h, w) = src.shape[:-2], src.shape[-2:]
    N = src.numel() // (h * w)
    dst = torch.empty(tuple(lead) + (H, W), dtype=torch.float32, device=src.device)
    check(lib().cf_pad2d(_f32(src), _f32(dst), N, h, w, y0, x0, H, W, _stream()), "cf_pad2d")
    return dst


def tta_accumulate(logits, acc, flip_h, flip_w, weight):
    B, K, H, W = logits.shape
    assert acc.shape == logits.shape
    check(lib().cf_tta_accumulate(_f32(logits), _f32(acc), B, K, H, W, int(flip_h), int(flip_w), float(weight), _stream()),
          "cf_tta_accumulate")
    return acc


def flip2d(src, flip_h, flip_w):
    H, W = src.shape[-2:]
    N = src.numel() // (H * W)
    dst = torch.empty_like(src)
    check(lib().cf_flip2d(_f32(src), _f32(dst), N, H, W, int(flip_h), int(flip_w), _stream()), "cf_flip2d")
    return dst


def tile_accumulate(pred, gauss, agg, cnt, lx, ly):
    K, ph, pw = pred.shape
    _, X, Y = agg.shape
    check(lib().cf_tile_accumulate(_f32(pred), _opt(gauss), _f32(agg), _f32(cnt), K, X, Y, lx, ly, ph, pw, _stream()),
          "cf_tile_accumulate")


def tile_finalize(agg, cnt):
    K, X, Y = agg.shape
    probs = torch.empty_like(agg)
    seg = torch.empty((X, Y), dtype=torch.uint8, device=agg.device)
    check(lib().cf_tile_finalize(_f32(agg), _f32(cnt), _f32(probs), _u8(seg), K, X, Y, _stream()), "cf_tile_finalize")
    return seg, probs


def tta_accumulate_3d(logits, acc, flip_d, flip_h, flip_w, weight):
    B, K, D, H, W = logits.shape
    assert acc.shape == logits.shape
    check(lib().cf_tta_accumulate_3d(_f32(logits), _f32(acc), B, K, D, H, W, int(flip_d), int(flip_h), int(flip_w), float(weight),
                                     _stream()), "cf_tta_accumulate_3d")
    return acc


def flip3d(src, flip_d, flip_h, flip_w):
    D, H, W = src.shape[-3:]
    N = src.numel() // (D * H * W)
    dst = torch.empty_like(src)
    check(lib().cf_flip3d(_f32(src), _f32(dst), N, D, H, W, int(flip_d), int(flip_h), int(flip_w), _stream()), "cf_flip3d")
    return dst


def tile_accumulate_3d(pred, gauss, agg, cnt, lx, ly, lz):
    K, px, py, pz = pred.shape
    _, X, Y, Z = agg.shape
    check(lib().cf_tile_accumulate_3d(_f32(pred), _opt(gauss), _f32(agg), _f32(cnt), K, X, Y, Z, lx, ly, lz, px, py, pz, _stream()),
          "cf_tile_accumulate_3d")


def argmax_channels(x):
    B, K = x.shape[0], x.shape[1]
    HW = x.numel() // (B * K)
    out = torch.empty((B,) + tuple(x.shape[2:]), dtype=torch.uint8, device=x.device)
    check(lib().cf_argmax_channels(_f32(x), _u8(out), B, K, HW, _stream()), "cf_argmax_channels")
    return out


def window_attention(qk, v, bias_table, heads, window, shift):
    """Swin windowed cross-attention core: qk [B,2C,H,W] (q | k of one input), v [B,C,H,W] (of the other), bias_table [(2w-1)^2, heads]
    -> [B,C,H,W] (nnunet/lib/swin_cross_attention.py:13-112 without the projections)."""
    B, C, H, W = v.shape
    assert qk.shape == (B, 2 * C, H, W) and bias_table.shape == ((2 * window - 1) ** 2, heads)
    out = torch.empty_like(v)
    check(lib().cf_window_attention(_f32(qk), _f32(v), _f32(bias_table), _f32(out), B, C, H, W, heads, window, shift, _stream()), "cf_window_attention")
    return out


def frame_boxes(x):
    """x [N,H,W] uint8 or float32 on the device -> int32 [N,4] = (x1, y1, x2, y2) of the non-zero pixels per frame, -1 for empty frames
    (torchvision.ops.masks_to_boxes as processor.py:140-160 uses it)."""
    assert x.is_cuda and x.dim() == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)
    N, H, W = x.shape
    boxes = torch.empty((N, 4), dtype=torch.int32, device=x.device)
    check(lib().cf_frame_boxes(x.data_ptr(), int(x.dtype == torch.float32), boxes.data_ptr(), N, H, W, _stream()), "cf_frame_boxes")
    return boxes


def sample_points(field, pts):
    """SpatialTransformerContour (integration.py:5-34): field [B,C,H,W], pts [B,2,P] (channel 0 along W, channel 1 along H) -> [B,C,P]."""
    B, C, H, W = field.shape
    assert pts.shape[0] == B and pts.shape[1] == 2
    P_ = pts.shape[2]
    out = torch.empty((B, C, P_), dtype=torch.float32, device=field.device)
    check(lib().cf_sample_points_2d(_f32(field), _f32(pts), _f32(out), B, C, H, W, P_, _stream()), "cf_sample_points_2d")
    return out


# ------------------------------------------------------------------------------------------------ export post-processing
def resize3d(src, new_shape, linear=(1, 1, 1)):
    """src [N,X,Y,Z] float32 -> [N,*new_shape]; per axis linear (order 1) or nearest (order 0); skimage 'edge' semantics."""
    N, X, Y, Z = src.shape
    X2, Y2, Z2 = (int(v) for v in new_shape)
    dst = torch.empty((N, X2, Y2, Z2), dtype=torch.float32, device=src.device)
    check(lib().cf_resize3d(_f32(src), _f32(dst), N, X, Y, Z, X2, Y2, Z2, int(linear[0]), int(linear[1]), int(linear[2]), _stream()),
          "cf_resize3d")
    return dst


def resample_data_or_seg(data, new_shape, is_seg, axis=None, order=3, do_separate_z=False, order_z=0):
    """nnunet/preprocessing/preprocessing.py:111-200 on the device for the orders the export uses (0 and 1).  data: numpy or
    device tensor (c, x, y, z); returns the same kind.  Segmentations of order 0 are nearest-neighbour in every axis; data is
    linear in-plane and `order_z` along the anisotropic axis when do_separate_z."""
    was_numpy = not torch.is_tensor(data)
    t = torch.from_numpy(np.ascontiguousarray(data)) if was_numpy else data
    dtype_in = t.dtype
    assert t.dim() == 4 and len(new_shape) == 3, "data must be (c, x, y, z)"
    if tuple(t.shape[1:]) == tuple(int(v) for v in new_shape):
        return data
    if order not in (0, 1) or order_z not in (0, 1):
        raise NotImplementedError("export resampling is built for interpolation orders 0 and 1 (got %s / %s)" % (order, order_z))
    if is_seg and order != 0:
        raise NotImplementedError("segmentation resampling is built for order 0")
    dev = torch.device("cuda", torch.cuda.current_device())
    lin = [int(order)] * 3
    if do_separate_z:
        assert len(axis) == 1, "only one anisotropic axis supported"
        lin[int(axis[0])] = int(order_z)
    out = resize3d(t.to(dev, dtype=torch.float32).contiguous(), new_shape, lin)
    if is_seg:
        out = out.round()
    out = out.to(dtype_in)
    return out.cpu().numpy() if was_numpy else out


def remove_all_but_the_largest_connected_component(image, for_which_classes, volume_per_voxel, minimum_valid_object_size=None):
    """nnunet/postprocessing/connected_components.py:51-107 on the device.  image: uint8 device tensor [Z,Y,X] or [Y,X]
    (modified in place); for_which_classes: ints or tuples of ints (joint regions) or None (all foreground labels).
    Returns (image, largest_removed, kept_size) like the reference."""
    import ctypes
    assert image.dtype == torch.uint8 and image.is_cuda and image.is_contiguous()
    shape = tuple(image.shape)
    D, H, W = (1,) * (3 - len(shape)) + shape
    n = image.numel()
    if for_which_classes is None:
        for_which_classes = [int(v) for v in torch.unique(image).tolist() if v > 0]
    assert 0 not in for_which_classes, "cannot remove background"
    labels = torch.empty(n, dtype=torch.int32, device=image.device)
    counts = torch.empty(n, dtype=torch.int32, device=image.device)
    changed = torch.zeros(1, dtype=torch.int32, device=image.device)
    largest_removed, kept_size = {}, {}
    for c in for_which_classes:
        key = tuple(c) if isinstance(c, (list, tuple)) else c
        vals = list(key) if isinstance(key, tuple) else [key]
        cls = (ctypes.c_uint8 * len(vals))(*[int(v) for v in vals])
        check(lib().cf_cc_init(_u8(image), labels.data_ptr(), n, ctypes.cast(cls, ctypes.c_void_p), len(vals), _stream()), "cf_cc_init")
        while True:
            changed.zero_()
            for _ in range(8):
                check(lib().cf_cc_sweep(labels.data_ptr(), D, H, W, changed.data_ptr(), _stream()), "cf_cc_sweep")
            if int(changed.item()) == 0:
                break
        counts.zero_()
        check(lib().cf_cc_count(labels.data_ptr(), counts.data_ptr(), n, _stream()), "cf_cc_count")
        sizes = counts[counts > 0]
        largest_removed[key] = None
        kept_size[key] = None
        if sizes.numel() > 0:
            max_count = int(sizes.max().item())
            kept_size[key] = max_count * volume_per_voxel
            thr = -1.0 if minimum_valid_object_size is None else float(minimum_valid_object_size[key])
            gone = sizes[sizes != max_count]
            if thr >= 0:
                gone = gone[gone.double() * volume_per_voxel < thr]
            if gone.numel() > 0:
                largest_removed[key] = int(gone.max().item()) * volume_per_voxel
            check(lib().cf_cc_remove(_u8(image), labels.data_ptr(), counts.data_ptr(), n, max_count, float(volume_per_voxel), thr, _stream()),
                  "cf_cc_remove")
    return image, largest_removed, kept_size
