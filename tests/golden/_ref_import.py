"""Import helper used ONLY by tests/golden/make_golden.py, in the build container.

It makes the reference's hot-path classes importable from /root/reference by
putting inert stand-ins in ``sys.modules`` for packages that are absent from
the image (timm, torchvision, monai, cv2, batchgenerators, ...) and for the
in-repo modules that the reference's .gitignore dropped (nnunet.lib.raft, ...).
The stand-ins only satisfy ``import`` statements; none of them is executed on
the paths the golden vectors exercise, except the three trivial timm helpers
re-stated below.

Nothing from /root/reference is copied: the reference is imported in place,
its outputs on seeded inputs are written to tests/golden/*.npz, and only those
arrays (data, not source) travel with the repo.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CINEFLOW_REFERENCE", "/root/reference")

_STUB_ROOTS = (
    "timm", "torchvision", "monai", "cv2", "batchgenerators", "matplotlib",
    "SimpleITK", "skimage", "nibabel", "medpy", "kornia", "pystrum", "fvcore",
    "tensorboard", "ruamel", "unittest2", "tqdm_missing",
)
_STUB_EXACT = tuple("nnunet.lib." + m for m in (
    "spacetimeAttention", "convlstm", "swin_cross_attention_old", "sfb", "raft",
    "raft_initial", "raft_extractor", "raft_extractor_seg", "gma",
    "swin_cross_attention_return", "loss", "vq_vae",
))


class _Anything:
    """Class returned for every attribute of a stub module."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed symbol was executed on the golden path")


class _StubModule(types.ModuleType):
    __path__ = []  # behave like a package so that submodule imports resolve

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        return type(name, (_Anything,), {})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if fullname in _STUB_EXACT:
            # only stub in-repo modules that are really missing from the snapshot
            rel = os.path.join(REFERENCE_ROOT, *fullname.split(".")) + ".py"
            if os.path.exists(rel):
                return None
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        if root in _STUB_ROOTS:
            try:
                # a real install wins over the stub
                for finder in sys.meta_path:
                    if finder is self:
                        continue
                    spec = finder.find_spec(fullname, path, target) if hasattr(finder, "find_spec") else None
                    if spec is not None:
                        return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        if module.__name__ == "timm.models.layers":
            _fill_timm_layers(module)


def _fill_timm_layers(mod):
    import torch
    from torch import nn

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):  # inference: identity
            return x

    mod.to_2tuple = to_2tuple
    mod.trunc_normal_ = trunc_normal_
    mod.DropPath = DropPath


def install():
    """Make `import nnunet...` resolve to the read-only reference tree."""
    sys.dont_write_bytecode = True
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch
    # SegFlowGaussian.py:350 passes torch.cuda.FloatTensor as a dtype tag
    if not hasattr(torch.cuda, "FloatTensor"):
        torch.cuda.FloatTensor = torch.FloatTensor


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nnunet"))
