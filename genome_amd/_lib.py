"""ctypes binding of the C-ABI library (include/genome_amd.h -> genome_amd/libgenome_amd.so).

This is the Python twin of the JNI stub shown in INTEGRATION.md: plain pointers and sizes, no torch
types.  There is no fallback: if the library is missing, or a call fails, an exception is raised.

Import order note: when a process also uses torch (bench.py for torch.distributed), `import torch`
must come BEFORE the first call into this module so that the HIP runtime torch bundles
(libamdhip64.so, SONAME libamdhip64.so.7) is the single runtime in the process.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GK_LIB_PATH: load another build of the SAME library — the TEST build (libgenome_amd_test.so: the product's objects plus the test
# hooks; tests/conftest.py selects it), a -DGK_TIMERS or tuning variant under genome_amd/variants/ — without overwriting the
# product .so; there is still no fallback if the file is missing.
LIB_PATH = os.environ.get("GK_LIB_PATH") or os.path.join(_HERE, "libgenome_amd.so")

GK_OK = 0
GK_E_INVALID, GK_E_KLEN, GK_E_UNSUPPORTED_K, GK_E_CAPACITY = -1, -2, -3, -4
GK_E_HIP, GK_E_NODEVICE, GK_E_FORMAT, GK_E_STATE, GK_E_COMM = -5, -6, -7, -8, -9
_NAMES = {-1: "GK_E_INVALID", -2: "GK_E_KLEN", -3: "GK_E_UNSUPPORTED_K", -4: "GK_E_CAPACITY", -5: "GK_E_HIP",
          -6: "GK_E_NODEVICE", -7: "GK_E_FORMAT", -8: "GK_E_STATE", -9: "GK_E_COMM"}


class GkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_NAMES.get(code, code)}: {msg}")
        self.code = code


class KeyLengthError(GkError, AssertionError):
    """The reference raises AssertionError (`assert(key.length == k)`, ArrayDNAMap.scala:182)."""


_lib = None

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p

# every symbol include/genome_amd.h declares: (restype, argtypes)
SIGNATURES = {
    "gk_device_count": (C.c_int, []),
    "gk_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "gk_ctx_destroy": (None, [vp]),
    "gk_last_error": (C.c_char_p, [vp]),
    "gk_ctx_device": (C.c_int, [vp]),
    "gk_ctx_trim": (C.c_int, [vp]),
    "gk_ctx_sync": (C.c_int, [vp]),
    "gk_ctx_mem_stats": (C.c_int, [vp, u64p, u64p, u64p, C.c_int]),
    "gk_ctx_set_mem_budget": (C.c_int, [vp, C.c_uint64]),
    "gk_map_add_map": (C.c_int, [vp, vp]),
    "gk_map_verify": (C.c_int, [vp, u64p, u64p, u64p, u64p]),
    "gk_map_set_max_batch_keys": (C.c_int, [vp, C.c_uint64]),
    "gk_map_trim": (C.c_int, [vp]),
    "gk_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "gk_host_free": (C.c_int, [vp, vp]),
    "gk_host_register": (C.c_int, [vp, vp, C.c_size_t]),
    "gk_host_unregister": (C.c_int, [vp, vp]),
    "gk_dev_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "gk_dev_free": (C.c_int, [vp, vp]),
    "gk_dev_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "gk_dev_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "gk_dev_stream_bench": (C.c_int, [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "gk_map_create": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "gk_map_create_for_graph": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "gk_map_destroy": (None, [vp]),
    "gk_map_k": (C.c_int, [vp]),
    "gk_map_clear": (C.c_int, [vp]),
    "gk_map_set_insert_path": (C.c_int, [vp, C.c_int]),
    "gk_map_size": (C.c_int, [vp, u64p]),
    "gk_map_slots": (C.c_int, [vp, u64p]),
    "gk_map_count_reads": (C.c_int, [vp, u8p, C.c_size_t, C.c_uint64, u64p]),
    "gk_map_prefetch_reads": (C.c_int, [vp, u8p, C.c_size_t, C.c_uint64]),
    "gk_map_count_reads_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, u64p]),
    "gk_map_update_inc": (C.c_int, [vp, u64p, u64p, C.c_uint64]),
    "gk_map_update_inc_dev": (C.c_int, [vp, vp, C.c_uint64]),
    "gk_map_add_counts": (C.c_int, [vp, u64p, u64p, i32p, C.c_uint64]),
    "gk_map_filter_lt": (C.c_int, [vp, C.c_int32]),
    "gk_map_get_batch": (C.c_int, [vp, u64p, u64p, C.c_uint64, i32p, u8p]),
    "gk_map_export": (C.c_int, [vp, u64p, u64p, i32p, C.c_uint64, u64p]),
    "gk_map_stats": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "gk_map_last_count_kernel": (C.c_int, [vp, C.POINTER(C.c_float), u64p]),
    "gk_map_last_phase_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "gk_shard_reads_dev": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.c_int, C.c_int, vp, C.c_uint64, u64p]),
    "gk_map_count_foreign": (C.c_int, [vp, C.c_int, C.c_int, u64p]),
    "gk_owner_of": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int]),
    "gk_skm_slot_bytes": (C.c_int, [C.c_int]),
    "gk_shard_superkmers_dev": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.c_int, C.c_int, vp, C.c_uint64, u64p, u64p]),
    "gk_map_count_superkmers_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u64p]),
    "gk_dist_unique_id": (C.c_int, [vp]),
    "gk_dist_create": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "gk_dist_destroy": (None, [vp]),
    "gk_dist_rank": (C.c_int, [vp]),
    "gk_dist_world": (C.c_int, [vp]),
    "gk_dist_barrier": (C.c_int, [vp]),
    "gk_dist_allreduce_f64": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "gk_dist_count_reads_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, u64p, u64p]),
    "gk_dist_route_begin": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.c_int]),
    "gk_dist_count_routed": (C.c_int, [vp, vp, u64p, u64p]),
    "gk_dist_last_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "gk_dist_size": (C.c_int, [vp, vp, u64p]),
    "gk_dist_gather_map": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "gk_dist_gather_classified_map": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "gk_dist_classify_queries": (C.c_int, [vp, u64p]),
    "gk_vmap_create": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "gk_vmap_destroy": (None, [vp]),
    "gk_vmap_k": (C.c_int, [vp]),
    "gk_vmap_size": (C.c_int, [vp, u64p]),
    "gk_vmap_put_new_batch": (C.c_int, [vp, u64p, u64p, u64p, C.c_uint64]),
    "gk_vmap_update_batch": (C.c_int, [vp, u64p, u64p, u64p, C.c_uint64]),
    "gk_vmap_get_all_batch": (C.c_int, [vp, u64p, u64p, C.c_uint64, u64p, u64p, C.c_uint64, u64p]),
    "gk_vmap_get_batch": (C.c_int, [vp, u64p, u64p, C.c_uint64, u64p, u8p]),
    "gk_vmap_export": (C.c_int, [vp, u64p, u64p, u64p, C.c_uint64, u64p]),
    "gk_graph_position_map": (C.c_int, [vp, vp, u64p]),
    "gk_graph_node_lookup": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gk_graph_nodes_by_id": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_uint64, u64p, u64p, u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gk_graph_edges_by_id": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64p, u8p, u8p]),
    "gk_graph_add_node": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]),
    "gk_graph_replace_start": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "gk_graph_id_bounds": (C.c_int, [vp, u64p, u64p]),
    "gk_graph_remove_edges_by_id": (C.c_int, [vp, u32p, C.c_uint64, u64p]),
    "gk_support_create": (C.c_int, [vp, C.POINTER(vp)]),
    "gk_support_destroy": (None, [vp]),
    "gk_support_size": (C.c_int, [vp, u64p, u64p, u64p]),
    "gk_support_last_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "gk_support_export": (C.c_int, [vp, u32p, u32p, u32p, C.c_uint64, u64p]),
    "gk_graph_walk_pairs": (C.c_int, [vp, vp, vp, u8p, C.c_size_t, C.c_uint64, C.c_int, C.c_int]),
    "gk_graph_split_by_support": (C.c_int, [vp, vp, C.c_int, u64p, u64p]),
    "gk_graph_replace_end": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "gk_graph_build": (C.c_int, [vp, C.POINTER(vp)]),
    "gk_graph_destroy": (None, [vp]),
    "gk_graph_counts": (C.c_int, [vp, u64p, u64p, u64p]),
    "gk_graph_simplify": (C.c_int, [vp]),
    "gk_graph_remove_bubbles": (C.c_int, [vp]),
    "gk_graph_remove_edges": (C.c_int, [vp, u64p, u64p, u8p, C.c_uint64, u64p]),
    "gk_graph_retain_largest": (C.c_int, [vp, u64p, u64p]),
    "gk_graph_component_stats": (C.c_int, [vp, C.POINTER(C.c_uint32), u64p, C.c_uint64, u64p]),
    "gk_graph_checksum": (C.c_int, [vp, u64p, u64p]),
    "gk_graph_bucketed_table_stats": (C.c_int, [vp, C.POINTER(C.c_float), u64p]),
    "gk_graph_classified_by_owners": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "gk_graph_build_stats": (C.c_int, [vp, C.POINTER(C.c_float), u64p, C.POINTER(C.c_int)]),
    "gk_graph_export_nodes": (C.c_int, [vp, u64p, u64p, C.c_uint64, u64p]),
    "gk_graph_export_edges": (C.c_int, [vp, u64p, u64p, u64p, u64p, i64p, i64p, C.c_uint64, u64p, u8p, C.c_uint64, u64p]),
    "gk_graph_out_order": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gk_synth_reads_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "gk_prefilter_create": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "gk_prefilter_destroy": (None, [vp]),
    "gk_prefilter_add_reads": (C.c_int, [vp, u8p, C.c_size_t, C.c_uint64]),
    "gk_prefilter_add_reads_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
    "gk_map_count_reads_prefiltered": (C.c_int, [vp, vp, u8p, C.c_size_t, C.c_uint64, u64p, u64p]),
    "gk_map_count_reads_prefiltered_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, u64p, u64p]),
    "gk_prefilter_stats": (C.c_int, [vp, u64p, u64p, u64p, u64p]),
}


# what only the TEST build of the library exports (include/genome_amd_test.h; genome_amd/libgenome_amd_test.so)
TEST_SIGNATURES = {
    "gk_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "gk_dist_create_loopback": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
}
TEST_LIB_PATH = os.path.join(_HERE, "libgenome_amd_test.so")


def lib():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(make -C genome_amd/csrc). genome_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        for name, (res, args) in TEST_SIGNATURES.items():     # present in the test build only
            f = getattr(L, name, None)
            if f is not None:
                f.restype = res
                f.argtypes = args
        _lib = L
    return _lib


def test_hook(name: str):
    """an entry point of the test build; a clear error when the product library is the one that is loaded"""
    f = getattr(lib(), name, None)
    if f is None:
        raise GkError(GK_E_STATE, f"{name} is a test hook: it is not in {os.path.basename(LIB_PATH)} — load the test build "
                                  f"(GK_LIB_PATH={TEST_LIB_PATH})")
    return f


def device_count() -> int:
    return lib().gk_device_count()


def check(rc: int, ctx=None):
    if rc == GK_OK:
        return
    msg = lib().gk_last_error(ctx)
    msg = msg.decode() if msg else ""
    if rc == GK_E_KLEN:
        raise KeyLengthError(rc, msg)
    raise GkError(rc, msg)


def ptr(a: np.ndarray | None, ctype):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


def as_u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)
