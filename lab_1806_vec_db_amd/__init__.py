"""lab_1806_vec_db_amd -- MI355X-native distance / top-k engine behind lab-1806-vec-db's index traits.

Only the hot path lives here: csrc/ (hand-written gfx950 HIP kernels + the C ABI, include/vdbhip.h) and
the host-side mirror of the reference's index / VecDB surface.  There is no CPU fallback.
"""
from ._lib import COSINE, L2SQR, VdbError  # noqa: F401
from .index import GpuIndex, calc_dist, calc_dist_u8, merge_topk  # noqa: F401
from .vecdb import VecDB  # noqa: F401

__all__ = ["GpuIndex", "VecDB", "calc_dist", "calc_dist_u8", "merge_topk", "VdbError", "L2SQR", "COSINE"]
