"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/cineflow.h declares, the ctypes
signatures match the header, the host logic (steps, Gaussian, padding, crop arithmetic, chunk order, state_dict
layout, sharding) matches the oracle, and the product refuses to run without its HIP library."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cineflow.h")


def _header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(cf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds.append("P")
                elif re.match(r"(const\s+)?long\b", a):
                    kinds.append("L")
                elif re.match(r"(const\s+)?float\b", a):
                    kinds.append("F")
                elif re.match(r"(const\s+)?double\b", a):
                    kinds.append("D")
                elif re.match(r"(const\s+)?int\b", a):
                    kinds.append("I")
                else:
                    raise AssertionError("unparsed argument %r of %s" % (a, name))
        protos[name] = kinds
    return protos


def test_library_exports_every_declared_symbol():
    from cineflow import _lib
    h = _lib.lib()
    protos = _header_prototypes()
    assert len(protos) >= 25
    for name in list(protos) + ["cf_last_error"]:
        assert hasattr(h, name), "symbol %s missing from libcineflow_hip.so" % name
    assert h.cf_version() >= 100


def test_ctypes_signatures_match_header():
    from cineflow import _lib
    protos = _header_prototypes()
    protos.pop("cf_version")
    kind = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_long: "L", ctypes.c_float: "F", ctypes.c_double: "D"}
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, argtypes in _lib.SIGNATURES.items():
        assert [kind[a] for a in argtypes] == protos[name], name


def test_argument_validation_needs_no_gpu():
    """Shape errors are caught on the host before any launch (returns CF_ERR_ARG with a message)."""
    from cineflow import _lib
    h = _lib.lib()
    rc = h.cf_warp_bilinear_2d(None, None, None, 1, 1, 8, 8, None)
    assert rc == -1 and b"null pointer" in h.cf_last_error()
    rc = h.cf_attention_cf(1, 0, 1, 0, 1, 0, 1, 1, 4, 8, 33, 32, None)
    assert rc == -1 and b"multiples of 32" in h.cf_last_error()
    rc = h.cf_group_norm(1, None, None, None, 1, 2, 12, 16, 8, 1e-5, 0, 0, 1, None)
    assert rc == -1


def test_product_fails_loudly_without_library(monkeypatch):
    from cineflow import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcineflow_hip.so")
    with pytest.raises(_lib.CineflowLibraryError):
        _lib.lib()


def test_ops_reject_cpu_tensors():
    from cineflow import ops
    with pytest.raises(TypeError):
        ops.warp_bilinear(torch.zeros(1, 2, 8, 8), torch.zeros(1, 1, 8, 8))


# ---------------------------------------------------------------- host logic vs oracle
def test_host_steps_gaussian_padding_match_oracle():
    from cineflow import inference as inf
    from oracle import ops as OO
    rng = np.random.RandomState(1)
    for _ in range(500):
        dim = rng.choice((2, 3))
        patch = tuple(int(v) for v in rng.randint(16, 512, dim))
        image = tuple(max(int(rng.randint(p // 2, p * 6)), p) for p in patch)
        step = float(rng.uniform(0.05, 1))
        assert inf.compute_steps_for_sliding_window(patch, image, step) == OO.compute_steps_for_sliding_window(patch, image, step)
    assert inf.compute_steps_for_sliding_window((64, 130), (128, 260), 0.5) == [[0, 32, 64], [0, 65, 130]]
    assert np.array_equal(inf.get_gaussian((64, 48)), OO.get_gaussian((64, 48)))
    x = rng.randn(3, 50, 61).astype(np.float32)
    a, sa = inf.pad_nd_image(x, (64, 64), "constant", {"constant_values": 0}, True)
    b, sb = OO.pad_nd_image(x, (64, 64), "constant", {"constant_values": 0}, True)
    assert np.array_equal(a, b) and sa == sb
    a = inf.pad_nd_image(x, (32, 32))
    assert a is x


def test_processor_arithmetic_matches_oracle():
    from cineflow.inference import Processor
    from oracle.models import Processor as OP
    for crop, image in ((128, 224), (192, 384), (16, 40)):
        p, o = Processor(crop, image), OP(crop, image)
        rng = np.random.RandomState(2)
        for _ in range(200):
            c = (int(rng.randint(0, image)), int(rng.randint(0, image)))
            a, b = p.adjust_cropping_window(c), o.adjust_cropping_window(c)
            assert a["crop_indices"] == b["crop_indices"]
            assert a["padding_need"] == b["padding_need"].tolist()


def test_chunk_orders_match_torch_chunk():
    from cineflow.inference import chunk_orders
    for T in (2, 3, 4, 5, 12, 30, 31):
        idx = torch.arange(1, T)
        c1, c2 = (torch.chunk(idx, 2) + (torch.tensor([], dtype=torch.long),))[:2] if T > 2 else (idx, torch.tensor([], dtype=torch.long))
        want1 = [0] + c1.tolist()
        want2 = [0] + torch.flip(c2, dims=[0]).tolist()
        assert chunk_orders(T) == (want1, want2), T


def _shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_layout_matches_oracle_and_reference_names():
    """The product modules declare exactly the reference's state_dict keys and shapes (via the oracle, whose
    strict load against the reference is checked in make_golden.py)."""
    from cineflow import models as PM
    from cineflow import nn as PN
    from oracle import models as OM
    red = dict(in_dims=[6, 16, 32], out_encoder_dims=[8, 16, 32])
    pairs = [
        (PM.SegFlowGaussian(image_size=64, d_model=32, dim_feedforward=64, motion_appearance=True, **red),
         OM.SegFlowGaussian(image_size=64, d_model=32, dim_feedforward=64, motion_appearance=True, **red)),
        (PM.SegFlowGaussian(image_size=64, d_model=32, dim_feedforward=48, motion_appearance=False, **red),
         OM.SegFlowGaussian(image_size=64, d_model=32, dim_feedforward=48, motion_appearance=False, **red)),
        (PM.SegFlowGaussian(image_size=256, motion_appearance=False, dim_feedforward=2048, raft=True),
         OM.SegFlowGaussian(image_size=256, motion_appearance=False, dim_feedforward=2048, raft=True)),
        (PM.ModelWrap(PM.OpticalFlowModelSuccessive(64, 1, **red), PM.OpticalFlowModelSuccessive(64, 6, **red)),
         OM.ModelWrap(OM.OpticalFlowModelSuccessive(64, 1, **red), OM.OpticalFlowModelSuccessive(64, 6, **red))),
        (PM.Generic_UNet(1, 8, 4, 3), OM.GenericUNet2D(1, 8, 4, 3)),
        (PM.Generic_UNet(1, 32, 4, 6), OM.GenericUNet2D(1, 32, 4, 6)),
        (PN.ConvGRUCell((8, 8), 32, 32), OM.ConvGRUCell((8, 8), 32, 32)),
    ]
    for prod, ora in pairs:
        assert prod.state_shapes() == _shapes(ora), type(prod).__name__


def test_full_width_parameter_counts():
    """SURVEY.md appendix A [measured]: 25 357 698 params for the video.yaml model, 25 267 906 for raft_config.yaml."""
    from cineflow import models as PM
    def count(m):
        return sum(int(np.prod(s)) for k, s in m.state_shapes().items() if not k.endswith("grid"))
    assert count(PM.SegFlowGaussian(image_size=256, motion_appearance=False, dim_feedforward=2048)) == 25357698
    assert count(PM.SegFlowGaussian(image_size=256, motion_appearance=True, dim_feedforward=3072)) == 25267906


def test_seeded_weights_are_deterministic_and_name_keyed():
    from cineflow.weights import seeded_state_dict
    a = seeded_state_dict({"x.conv1.weight": (4, 3, 3, 3), "x.norm1.weight": (4,), "m.grid": (1, 2, 4, 4)}, seed=3)
    b = seeded_state_dict({"x.norm1.weight": (4,), "x.conv1.weight": (4, 3, 3, 3)}, seed=3)
    assert "m.grid" not in a
    assert torch.equal(a["x.conv1.weight"], b["x.conv1.weight"]) and torch.equal(a["x.norm1.weight"], b["x.norm1.weight"])
    c = seeded_state_dict({"x.conv1.weight": (4, 3, 3, 3)}, seed=4)
    assert not torch.equal(a["x.conv1.weight"], c["x.conv1.weight"])
    assert abs(float(a["x.norm1.weight"].mean()) - 1.0) < 0.3


def test_shard_is_the_reference_partition():
    from cineflow.parallel import shard
    items = list(range(11))
    parts = [shard(items, r, 4) for r in range(4)]
    assert parts[1] == items[1::4]
    assert sorted(sum(parts, [])) == items


def test_properties_pkl_reader_rebuilds_plain_values_and_refuses_globals(tmp_path):
    """the voxelmorph_saver consumer reads <pkl_path>/<case>.pkl files another pipeline may have written -- nnU-Net property
    dicts (OrderedDict, lists, tuples, numpy arrays / scalars) load, any other global is refused before it is imported or called."""
    import collections
    import pickle
    from cineflow.voxelmorph_saver import load_plain_pickle
    props = collections.OrderedDict(original_size_of_raw_data=np.array([10, 256, 216]), original_spacing=np.array([10.0, 1.25, 1.25]),
                                    list_of_data_files=["a_0000.nii.gz"], itk_spacing=(1.25, 1.25, 10.0), crop_bbox=[[0, 10], [0, 256], [0, 216]],
                                    classes=np.array([0, 1, 2, 3], dtype=np.int16), size_after_cropping=(10, 256, 216), use_nonzero_mask_for_norm={0: False},
                                    padding_need=np.arange(40, dtype=np.int64).reshape(4, 10), voxelmorph_size_before=[256, 216, 10],
                                    a_scalar=np.float64(1.5), an_int_scalar=np.int32(7))
    ok = tmp_path / "ok.pkl"
    with open(ok, "wb") as f:
        pickle.dump(props, f)
    got = load_plain_pickle(str(ok))
    assert list(got.keys()) == list(props.keys())
    assert np.array_equal(got["padding_need"], props["padding_need"]) and got["a_scalar"] == 1.5 and got["itk_spacing"] == (1.25, 1.25, 10.0)

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    bad = tmp_path / "bad.pkl"
    with open(bad, "wb") as f:
        pickle.dump({"padding_need": np.zeros((4, 1)), "x": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        load_plain_pickle(str(bad))
    assert not (tmp_path / "pwned").exists()


def test_conv_f16s_output_sample_limit_falls_back_to_fp32_kernel():
    """a destination buffer whose sample exceeds the epilogue's 32-bit store range makes f16s_dynamic_ok say no (the layer then
    takes the exact fp32 kernel) instead of raising from inside the library.  Host logic only."""
    from cineflow import ops

    class Fake:
        def __init__(self, shape):
            self.shape = shape

        def numel(self):
            n = 1
            for v in self.shape:
                n *= v
            return n
    x = Fake((1, 16, 4096, 4096))                                            # 1 GiB input sample: fine (< 2 GiB)
    assert ops.f16s_dynamic_ok(x, None, 3, out_sample_elems=8 * 4096 * 4096, out_hw=4096 * 4096)
    assert not ops.f16s_dynamic_ok(x, None, 3, out_sample_elems=16 * 4096 * 4096, out_hw=4096 * 4096)      # 1 GiB output sample
    assert not ops.f16s_dynamic_ok(Fake((1, 8, 16, 16)), None, 3, out_sample_elems=2 ** 25, out_hw=256)      # 8 images per workgroup
