"""ctypes binding of libvdbhip.so (C ABI: include/vdbhip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises on import
of the symbol table, and every compute call fails when no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDBHIP_LIB") or os.path.join(_HERE, "libvdbhip.so")  # override: measurement builds

L2SQR, COSINE = 0, 1

f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
intp = C.POINTER(C.c_int)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
u64 = C.c_uint64

# name -> argtypes; every function returns int status (except the two noted below)
SIGNATURES = {
    "vdb_device_count": [intp],
    "vdb_index_create": [C.c_int, u64, C.c_int, C.POINTER(vp)],
    "vdb_index_create_u8": [C.c_int, u64, C.c_int, C.POINTER(vp)],
    "vdb_index_row_u8": [vp, u64, u8p],
    "vdb_index_is_u8": [vp, intp],
    "vdb_index_destroy": [vp],
    "vdb_index_len": [vp, u64p],
    "vdb_index_dim": [vp, u64p],
    "vdb_index_dist": [vp, intp],
    "vdb_index_row": [vp, u64, f32p],
    "vdb_index_add": [vp, f32p, u64, u64p],
    "vdb_index_add_device": [vp, vp, u64, u64p],
    "vdb_index_swap_remove": [vp, u64],
    "vdb_index_set_id_offset": [vp, u64],
    "vdb_calc_dist": [C.c_int, f32p, f32p, u64, C.c_int, f32p],
    "vdb_flat_knn": [vp, f32p, u64, u64, u64, u64p, f32p, u64p],
    "vdb_flat_knn_device": [vp, vp, u64, u64, u64, vp, vp, vp, vp],
    "vdb_flat_knn_device_begin": [vp, vp, u64, u64, u64, vp, vp, vp, vp, C.POINTER(vp)],
    "vdb_flat_knn_device_end": [vp],
    "vdb_flat_shortlist_keys": [vp, f32p, u64, u64, C.c_int, f32p, f32p, f32p, f32p],
    "vdb_flat_set_mode": [vp, C.c_int],
    "vdb_index_prepare": [vp, C.c_int],
    "vdb_flat_fallback_count": [vp, u64p],
    "vdb_get_stat": [vp, C.c_char_p, u64p],
    "vdb_set_param": [vp, C.c_char_p, C.c_int64],
    "vdb_pq_attach": [vp, u64, u64, f32p, u8p],
    "vdb_pq_build": [vp, u64, u64, u64, u64, C.c_float, u64],
    "vdb_pq_clear": [vp],
    "vdb_pq_has": [vp, intp],
    "vdb_pq_info": [vp, u64p, u64p, u64p],
    "vdb_pq_export": [vp, f32p, u8p],
    "vdb_pq_create_lookup": [vp, f32p, u64, u64, f32p, f32p],
    "vdb_pq_adc_all": [vp, f32p, u64, u64, f32p],
    "vdb_flat_knn_pq": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_flat_knn_pq_device": [vp, vp, u64, u64, u64, u64, vp, vp, vp, vp],
    "vdb_hnsw_knn_device": [vp, vp, u64, u64, u64, u64, C.c_int, vp, vp, vp, vp],
    "vdb_flat_knn_pq_shard": [vp, f32p, u64, u64, u64, u64, u64p, u64p],
    "vdb_flat_knn_pq_shard_device": [vp, vp, u64, u64, u64, u64, vp, vp, vp],
    "vdb_pq_merge_resort": [u64p, u64p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_pq_merge_resort_device": [vp, vp, vp, u64, u64, u64, u64, vp, vp, vp, vp],
    "vdb_calc_dist_u8": [C.c_int, u8p, u8p, u64, C.c_int, C.POINTER(C.c_float)],
    "vdb_index_add_u8": [vp, u8p, u64, u64p],
    "vdb_flat_knn_u8": [vp, u8p, u64, u64, u64, u64p, f32p, u64p],
    "vdb_ivf_build": [vp, u64, u64, u64, C.c_float, u64],
    "vdb_ivf_attach": [vp, u64, f32p, u64p],
    "vdb_ivf_clear": [vp],
    "vdb_ivf_info": [vp, intp, u64p, u64p],
    "vdb_ivf_export": [vp, f32p, u64p],
    "vdb_ivf_knn": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_ivf_knn_device": [vp, vp, u64, u64, u64, u64, vp, vp, vp, vp],
    "vdb_hnsw_build": [vp, u64, u64, u64, u64, C.c_int],
    "vdb_hnsw_attach": [vp, u64, u64, u32p, u64p, u64p, u32p, u64p, C.c_int, u64, u64],
    "vdb_hnsw_clear": [vp],
    "vdb_hnsw_has": [vp, intp],
    "vdb_hnsw_info": [vp, u64p, u64p, u64p, intp, u64p, u64p, u64p],
    "vdb_hnsw_export": [vp, u32p, u64p, u64p, u32p, u64p],
    "vdb_hnsw_knn": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_hnsw_knn_pq": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_hnsw_last_stats": [vp, u64p, u64p],
    "vdb_merge_topk": [f32p, u64p, u64p, u64, u64, u64, u64p, f32p, u64p],
    "vdb_merge_topk_device": [vp, vp, vp, vp, u64, u64, u64, vp, vp, vp, vp],
    "vdb_merge_topk_gathered": [vp, vp, u64, u64, u64, u64, u64, u64, u64, vp, vp, vp, vp],
    "vdb_merge_topk_gathered_async": [vp, vp, u64, u64, u64, u64, u64, u64, u64, vp, vp, vp, vp],
    "vdb_ctx_create": [intp, C.c_int, C.POINTER(vp)],
    "vdb_ctx_unique_id": [vp, u64],
    "vdb_ctx_create_rank": [C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)],
    "vdb_ctx_destroy": [vp],
    "vdb_ctx_info": [vp, intp, intp, intp, intp],
    "vdb_sharded_create": [vp, u64, C.c_int, C.POINTER(vp)],
    "vdb_sharded_destroy": [vp],
    "vdb_sharded_set_rows": [vp, f32p, u64],
    "vdb_sharded_len": [vp, u64p],
    "vdb_sharded_local": [vp, C.c_int, C.POINTER(vp)],
    "vdb_sharded_flat_knn": [vp, f32p, u64, u64, u64, u64p, f32p, u64p],
    "vdb_sharded_pq_attach": [vp, u64, u64, f32p],
    "vdb_sharded_knn_pq": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_sharded_set_rows_replica": [vp, f32p, u64],
    "vdb_sharded_layout": [vp, intp],
    "vdb_replica_query_block": [u64, u64, u64, u64p, u64p],
    "vdb_sharded_hnsw_build": [vp, u64, u64, u64, u64, C.c_int],
    "vdb_sharded_hnsw_attach": [vp, u64, u64, u32p, u64p, u64p, u32p, u64p, C.c_int, u64, u64],
    "vdb_sharded_hnsw_knn": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_sharded_hnsw_knn_pq": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_sharded_poisoned": [vp, intp],
    "vdb_sharded_ivf_build": [vp, u64, u64, u64, C.c_float, u64],
    "vdb_sharded_ivf_knn": [vp, f32p, u64, u64, u64, u64, u64p, f32p, u64p],
    "vdb_stream_probe": [C.c_int, u64, C.c_int, f64p],
    "vdb_stream_probe_rows": [C.c_int, u64, C.c_int, C.c_uint32, f64p],
    "vdb_mfma_probe": [C.c_int, C.c_int, C.c_int, f64p, f64p],
    "vdb_mfma_probe_i8": [C.c_int, C.c_int, C.c_int, f64p, f64p],
    "vdb_latency_probe": [C.c_int, C.c_uint64, C.c_uint32, f64p],
    "vdb_fold_probe": [C.c_int, C.c_uint32, f64p],
    "vdb_prof_enable": [vp, C.c_int],
    "vdb_prof_reset": [vp],
    "vdb_prof_get": [vp, C.c_char_p, f64p, u64p, f64p],
}

_lib = None


class VdbError(RuntimeError):
    """Recoverable failure reported by libvdbhip (the reference raises PyRuntimeError, pyo3/mod.rs:65,85)."""


def _preload_torch_runtime():
    """PyTorch-ROCm wheels bundle their own libhsa-runtime64 / libamdhip64, and two HSA runtimes do not coexist in
    one process: whichever initialises second sees no GPUs ("No HIP GPUs are available").  Importing torch first
    (library load only, no GPU initialisation) lets libvdbhip.so's libhsa-runtime64.so.1 dependency resolve to the
    copy torch already mapped, so the process has a single runtime whatever the later call order is.  Without
    torch installed nothing happens -- the library itself does not need it."""
    import importlib.util

    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load():
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C lab_1806_vec_db_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = args
        fn.restype = C.c_int
    lib.vdb_last_error.restype = C.c_char_p
    lib.vdb_last_error.argtypes = []
    lib.vdb_version.restype = C.c_int
    lib.vdb_version.argtypes = []
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        msg = load().vdb_last_error().decode("utf-8", "replace")
        raise VdbError(f"libvdbhip error {status}: {msg}")
