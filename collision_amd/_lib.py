"""ctypes binding of libcollision_hip.so (C ABI: include/collision_hip.h).

The library is built in-tree by ``collision_amd/csrc/Makefile`` (``__graft_entry__.build()``).
If PyTorch is importable it is imported first so that this process has ONE HIP runtime:
torch bundles its own libamdhip64.so with the same SONAME, and whichever copy is loaded
first serves both (set COLLISION_AMD_NO_TORCH=1 to skip the import).
"""
import ctypes as C
import os
import sys
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# COLLISION_AMD_LIB: another build of the same ABI (A/B timing of two builds on one box: tools/ab_builds.sh)
LIB_PATH = Path(os.environ["COLLISION_AMD_LIB"]).resolve() if os.environ.get("COLLISION_AMD_LIB") else _HERE / "libcollision_hip.so"

c_void_pp = C.POINTER(C.c_void_p)

# name -> (restype or None for int status, argtypes)
_PROTOS = {
    "col_error_string": (C.c_char_p, [C.c_int]),
    "col_version": (C.c_int, []),
    "col_device_count": (None, [C.POINTER(C.c_int)]),
    "col_set_device": (None, [C.c_int]),
    "col_get_device": (None, [C.POINTER(C.c_int)]),
    "col_device_name": (None, [C.c_char_p, C.c_int]),
    "col_device_sync": (None, []),
    "col_malloc": (None, [c_void_pp, C.c_size_t]),
    "col_free": (None, [C.c_void_p]),
    "col_host_alloc": (None, [c_void_pp, C.c_size_t]),
    "col_host_free": (None, [C.c_void_p]),
    "col_memcpy_h2d": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "col_memcpy_d2h": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "col_memcpy_d2d": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "col_fill": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "col_stream_create": (None, [c_void_pp]),
    "col_stream_destroy": (None, [C.c_void_p]),
    "col_stream_sync": (None, [C.c_void_p]),
    "col_stream_wait_event": (None, [C.c_void_p, C.c_void_p]),
    "col_event_create": (None, [c_void_pp]),
    "col_event_destroy": (None, [C.c_void_p]),
    "col_event_record": (None, [C.c_void_p, C.c_void_p]),
    "col_event_sync": (None, [C.c_void_p]),
    "col_event_elapsed_ms": (None, [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    "col_reduce_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "col_reduce": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "col_reduce_list": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                               C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "col_morton": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "col_scan_scratch_bytes": (C.c_size_t, [C.c_uint64]),
    "col_scan_u32": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "col_local_scan": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "col_block_scan": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "col_radix_scratch_bytes": (C.c_size_t, [C.c_uint64, C.c_int, C.c_int]),
    "col_radix_sort": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                              C.c_int, C.c_void_p, C.c_int]),
    "col_radix_tile": (C.c_uint32, [C.c_uint64, C.c_int, C.c_int]),
    "col_radix_histogram": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "col_radix_scatter": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p]),
    "col_ref_block_sort": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint32,
                                  C.c_int, C.c_int, C.c_void_p]),
    "col_ref_scatter": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                               C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "col_bvh_build": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
    "col_bvh_refit": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                             C.c_int]),
    "col_traverse": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                            C.c_int]),
    "col_lbvh_scratch_bytes": (C.c_size_t, [C.c_uint32, C.c_int]),
    "col_lbvh": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_uint32, C.c_int]),
    "col_collide_scratch_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_int]),
    "col_radix_sort_msd": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                  C.c_void_p]),
    "col_collide_plan": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "col_minmax4_stage1_dev": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "col_collide_plan_partials": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]),
    "col_collide_plan_dev": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "col_collide": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_uint32]),
    "col_partition_scratch_bytes": (C.c_size_t, []),
    "col_partition_sample": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_int]),
    "col_partition_plan": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "col_partition_group": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint32, C.c_void_p, C.c_int]),
    "col_partition_unpack": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_int]),
    "col_unpack_radii": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]),
    "col_region_boxes": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint32, C.c_int]),
    "col_region_boxes_dev": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint32, C.c_int, C.c_void_p]),
    "col_select_overlap_multi": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]),
    "col_select_overlap_multi_dev": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "col_pack_slots": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int,
                              C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "col_ghost_scratch_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "col_traverse_ghost_slots": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]),
    "col_traverse_ghost_slots_dev": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "col_translate_pairs": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "col_traverse_chunked_scratch_bytes": (C.c_size_t, []),
    "col_traverse_chunked": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                    C.c_void_p]),
    "col_reduce_rtc_check": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "col_reduce_rtc_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "col_reduce_rtc_destroy": (C.c_int, [C.c_void_p]),
    "col_reduce_rtc": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_void_p]),
    "col_gather": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
    "col_scatter": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
    "col_find_offsets": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
}

# diagnostics entry points (include/collision_hip_debug.h): not part of the drop-in ABI; bound for tools/, bench.py's
# ablation legs and the tests that force a code path
_DEBUG_PROTOS = {
    "col_traverse_stats": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int,
                                  C.c_void_p, C.c_int]),
    "col_debug_xcc_census": (None, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "col_debug_traverse": (C.c_int, [C.c_int]),
    "col_debug_walk_profile": (C.c_int, [C.c_void_p, C.c_uint32]),
    "col_debug_lbvh": (C.c_int, [C.c_int]),
    "col_debug_leaf_blocks": (C.c_int, [C.c_float]),
    "col_debug_radix": (C.c_int, [C.c_int]),
    "col_debug_copy": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "col_debug_radix_stamps": (None, [C.c_void_p, C.c_int]),
    "col_debug_radix_tile": (None, [C.c_int]),
}

EXPORTS = tuple(_PROTOS)
DEBUG_EXPORTS = tuple(_DEBUG_PROTOS)
_PROTOS.update(_DEBUG_PROTOS)


class HipError(RuntimeError):
    """A C-ABI call returned a non-zero status (device errors surface here)."""


_cdll = None


def cdll():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _cdll
    if _cdll is not None:
        return _cdll
    if not LIB_PATH.exists():
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C collision_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
    if "torch" not in sys.modules and not os.environ.get("COLLISION_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401  (one HIP runtime per process, see module docstring)
        except Exception:
            pass
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int if restype is None else restype
    _cdll = lib
    return lib


def check(status):
    if status != 0:
        raise HipError("%s (status %d)" % (cdll().col_error_string(status).decode(), status))


class _Calls:
    """``call.col_xyz(args)``: status-returning entry points raise HipError on failure."""

    def __getattr__(self, name):
        restype, _ = _PROTOS[name]
        fn = getattr(cdll(), name)
        if restype is None:
            def wrapped(*args, _fn=fn):
                check(_fn(*args))
            wrapped.__name__ = name
            setattr(self, name, wrapped)
            return wrapped
        setattr(self, name, fn)
        return fn


call = _Calls()
