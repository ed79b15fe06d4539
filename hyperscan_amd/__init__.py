"""hyperscan_amd -- MI355X-native block-mode literal scan engine (HWLM layer of Hyperscan).

Host-side Python mirror of the reference's literal-matcher interface
(src/hwlm/hwlm_build.h, src/hwlm/hwlm.h) on top of the C ABI in include/hsgpu.h.
The compute path is hand-written HIP for gfx950 inside lib/libhsgpu.so; there is
no CPU fallback: importing works without a GPU, scanning raises without one.
"""
from .hwlm import (  # noqa: F401
    HWLM_ALL_GROUPS,
    HWLM_CONTINUE_MATCHING,
    HWLM_SUCCESS,
    HWLM_TERMINATE_MATCHING,
    HWLM_TERMINATED,
    HWLM_ERROR_UNKNOWN,
    HwlmLiteral,
    HwlmTable,
    Scratch,
    HsgpuError,
    hwlm_build,
    hwlm_exec,
    hwlm_size,
)
from ._native import lib_path, load_library  # noqa: F401

__all__ = [
    "HwlmLiteral", "HwlmTable", "Scratch", "HsgpuError", "hwlm_build", "hwlm_exec", "hwlm_size",
    "HWLM_ALL_GROUPS", "HWLM_CONTINUE_MATCHING", "HWLM_TERMINATE_MATCHING", "HWLM_SUCCESS",
    "HWLM_TERMINATED", "HWLM_ERROR_UNKNOWN", "lib_path", "load_library",
]
