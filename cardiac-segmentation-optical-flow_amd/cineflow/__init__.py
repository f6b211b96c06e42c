"""cineflow -- MI355X-native cardiac cine segmentation + optical-flow inference.

Host side (Python on PyTorch-ROCm for memory/streams) of the hot path; all
compute goes through the C-ABI HIP library ``libcineflow_hip.so`` (see
include/cineflow.h).  There is NO CPU fallback: calling an operator without the
library raises ``CineflowLibraryError``.
"""
__version__ = "0.1.0"
