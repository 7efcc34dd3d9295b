"""ctypes loader for rustfft_amd/lib/libmi355fft.so (the C ABI in include/mi355fft.h).

There is deliberately no fallback: if the shared library is missing it must be built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C rustfft_amd/csrc`), and if no gfx950 GPU
is visible every planning call fails with MI355FFT_ERR_NO_DEVICE.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355fft.so")

EXPORTS = [
    "mi355fft_device_count", "mi355fft_init", "mi355fft_plan_create", "mi355fft_plan_create_ex", "mi355fft_bluestein_inner_len", "mi355fft_plan_destroy", "mi355fft_plan_len",
    "mi355fft_plan_direction", "mi355fft_plan_precision", "mi355fft_plan_recipe_status", "mi355fft_scratch_len", "mi355fft_plan_describe",
    "mi355fft_process_inplace_host", "mi355fft_process_outofplace_host", "mi355fft_process_immutable_host",
    "mi355fft_process_inplace_dev", "mi355fft_process_outofplace_dev", "mi355fft_process_immutable_dev",
    "mi355fft_plan_num_kernels", "mi355fft_plan_kernel_name", "mi355fft_profile_inplace_dev",
    "mi355fft_multi_plan_create", "mi355fft_multi_plan_destroy", "mi355fft_multi_plan_shards", "mi355fft_multi_plan_device", "mi355fft_multi_plan_replica", "mi355fft_shard_rows",
    "mi355fft_multi_process_inplace_host", "mi355fft_multi_process_outofplace_host", "mi355fft_multi_process_immutable_host",
    "mi355fft_multi_process_inplace_dev", "mi355fft_multi_process_outofplace_dev", "mi355fft_multi_process_immutable_dev",
    "mi355fft_multi_synchronize", "mi355fft_device_cpulist", "mi355fft_multi_plan_shard_pinned", "mi355fft_multi_scatter_dev", "mi355fft_multi_gather_dev",
    "mi355fft_measure_copy_ceiling", "mi355fft_plan_set_fused", "mi355fft_plan_is_fused", "mi355fft_plan_fused_status", "mi355fft_plan_synchronize", "mi355fft_plan_set_fused_wait_limit", "mi355fft_plan_set_workspace_placement", "mi355fft_plan_set_chunk_batch", "mi355fft_plan_workspace_bytes", "mi355fft_plan_trim_workspaces", "mi355fft_strerror", "mi355fft_last_error", "mi355fft_version",
]


TWIDDLE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))


class RecipeNode(ctypes.Structure):
    """mi355fft_recipe_node (include/mi355fft.h): one node of the host planner's flattened Recipe tree."""
    _fields_ = [("kind", ctypes.c_int), ("left", ctypes.c_int), ("right", ctypes.c_int), ("len", ctypes.c_size_t)]


class PlanOptions(ctypes.Structure):
    """mi355fft_plan_options (include/mi355fft.h)."""
    _fields_ = [("struct_size", ctypes.c_size_t), ("algorithm", ctypes.c_int), ("twiddle_fn", TWIDDLE_FN), ("twiddle_ctx", ctypes.c_void_p),
                ("rader_inner_fft_data", ctypes.c_void_p), ("bluestein_twiddles", ctypes.c_void_p), ("bluestein_multiplier", ctypes.c_void_p),
                ("bluestein_inner_len", ctypes.c_size_t), ("recipe", ctypes.POINTER(RecipeNode)), ("recipe_nodes", ctypes.c_size_t)]


def bind(lib):
    """Attach argtypes/restypes for every symbol include/mi355fft.h declares."""
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.mi355fft_device_count.restype = ci
    lib.mi355fft_init.argtypes = [ci]
    lib.mi355fft_plan_create.argtypes = [sz, ci, ci, ctypes.POINTER(vp)]
    lib.mi355fft_plan_create_ex.argtypes = [sz, ci, ci, ctypes.POINTER(PlanOptions), ctypes.POINTER(vp)]
    lib.mi355fft_bluestein_inner_len.restype = sz
    lib.mi355fft_bluestein_inner_len.argtypes = [sz, ci]
    lib.mi355fft_plan_destroy.argtypes = [vp]
    lib.mi355fft_plan_len.restype = sz
    lib.mi355fft_plan_len.argtypes = [vp]
    lib.mi355fft_plan_direction.argtypes = [vp]
    lib.mi355fft_plan_precision.argtypes = [vp]
    lib.mi355fft_plan_recipe_status.argtypes = [vp]
    lib.mi355fft_scratch_len.restype = sz
    lib.mi355fft_scratch_len.argtypes = [vp, ci]
    lib.mi355fft_plan_describe.argtypes = [vp, ctypes.c_char_p, sz]
    lib.mi355fft_process_inplace_host.argtypes = [vp, vp, sz, vp, sz]
    lib.mi355fft_process_outofplace_host.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    lib.mi355fft_process_immutable_host.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    lib.mi355fft_process_inplace_dev.argtypes = [vp, vp, sz, vp]
    lib.mi355fft_process_outofplace_dev.argtypes = [vp, vp, vp, sz, vp]
    lib.mi355fft_process_immutable_dev.argtypes = [vp, vp, vp, sz, vp]
    pvp, pci = ctypes.POINTER(vp), ctypes.POINTER(ci)
    lib.mi355fft_multi_plan_create.argtypes = [sz, ci, ci, ctypes.POINTER(PlanOptions), pci, ci, ctypes.POINTER(vp)]
    lib.mi355fft_multi_plan_destroy.argtypes = [vp]
    lib.mi355fft_multi_plan_shards.argtypes = [vp]
    lib.mi355fft_multi_plan_device.argtypes = [vp, ci]
    lib.mi355fft_multi_plan_replica.restype = vp
    lib.mi355fft_multi_plan_replica.argtypes = [vp, ci]
    lib.mi355fft_shard_rows.argtypes = [sz, ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.mi355fft_multi_process_inplace_host.argtypes = [vp, vp, sz, vp, sz]
    lib.mi355fft_multi_process_outofplace_host.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    lib.mi355fft_multi_process_immutable_host.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    lib.mi355fft_multi_process_inplace_dev.argtypes = [vp, pvp, sz, pvp]
    lib.mi355fft_multi_process_outofplace_dev.argtypes = [vp, pvp, pvp, sz, pvp]
    lib.mi355fft_multi_process_immutable_dev.argtypes = [vp, pvp, pvp, sz, pvp]
    lib.mi355fft_multi_synchronize.argtypes = [vp, pvp]
    lib.mi355fft_multi_scatter_dev.argtypes = [vp, vp, ci, pvp, sz, pvp]
    lib.mi355fft_multi_gather_dev.argtypes = [vp, pvp, vp, ci, sz, pvp]
    lib.mi355fft_plan_num_kernels.argtypes = [vp]
    lib.mi355fft_plan_kernel_name.restype = ctypes.c_char_p
    lib.mi355fft_plan_kernel_name.argtypes = [vp, ci]
    lib.mi355fft_profile_inplace_dev.argtypes = [vp, vp, sz, vp, ci, ctypes.POINTER(ctypes.c_float), ci]
    lib.mi355fft_measure_copy_ceiling.argtypes = [sz, ctypes.POINTER(ctypes.c_double)]
    lib.mi355fft_plan_set_chunk_batch.argtypes = [vp, sz]
    lib.mi355fft_plan_set_workspace_placement.argtypes = [vp, ci]
    lib.mi355fft_plan_set_fused.argtypes = [vp, ci]
    lib.mi355fft_plan_is_fused.argtypes = [vp]
    lib.mi355fft_plan_fused_status.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint)]
    lib.mi355fft_plan_synchronize.argtypes = [vp, vp]
    # (entry points newer than some A/B builds kept in rustfft_amd/lib: bound where present; tests/test_cabi_cpu.py holds the shipped
    # library to the full EXPORTS list)
    if hasattr(lib, "mi355fft_plan_set_fused_wait_limit"):
        lib.mi355fft_plan_set_fused_wait_limit.argtypes = [vp, ci]
    if hasattr(lib, "mi355fft_device_cpulist"):
        lib.mi355fft_device_cpulist.argtypes = [ci, ctypes.c_char_p, sz]
        lib.mi355fft_multi_plan_shard_pinned.argtypes = [vp, ci]
    lib.mi355fft_plan_workspace_bytes.restype = sz
    lib.mi355fft_plan_workspace_bytes.argtypes = [vp]
    lib.mi355fft_plan_trim_workspaces.argtypes = [vp, ctypes.POINTER(sz)]
    lib.mi355fft_strerror.restype = ctypes.c_char_p
    lib.mi355fft_strerror.argtypes = [ci]
    lib.mi355fft_last_error.restype = ctypes.c_char_p
    lib.mi355fft_version.restype = ctypes.c_char_p
    return lib


_lib = None


def load(path=None):
    """Load the HIP library.  `path` exists only so tests can inject tests/emu's kernel-body emulator."""
    global _lib
    if path is not None:
        return bind(ctypes.CDLL(path))
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "rustfft_amd has no CPU fallback.")
        _lib = bind(ctypes.CDLL(LIB_PATH))
    return _lib
