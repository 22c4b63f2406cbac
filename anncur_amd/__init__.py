"""anncur_amd -- MI355X-native CUR nearest-neighbour path (drop-in for iesl/anncur's
eval/matrix_approx_zeshel.py + models/nearest_nbr.py flat search + the retrieval-eval loop).

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Compute: hand-written HIP kernels for gfx950 in anncur_amd/lib/libanncur_hip.so, bound via ctypes.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
