"""ctypes binding of libmvsn_hip.so (include/mvsn_hip.h).

There is no CPU fallback: if the library is missing or a call fails, the caller gets a
RuntimeError.  Tensors are handed over as raw device pointers plus sizes, and every call is
enqueued on torch's current HIP stream.
"""
import ctypes
import os
from ctypes import c_char_p, c_int, c_long, c_size_t, c_void_p, POINTER, Structure

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmvsn_hip.so")
_lib = None


class ConvDesc(Structure):
    """mvsn_conv_desc"""
    _fields_ = [("n", c_int), ("c_in", c_int), ("c_out", c_int), ("depth", c_int), ("rows", c_int),
                ("cols", c_int), ("kd", c_int), ("kh", c_int), ("kw", c_int), ("stride", c_int),
                ("dilation", c_int), ("precision", c_int)]


class ApplyJob(Structure):
    """mvsn_apply_job"""
    _fields_ = [("x", c_void_p), ("stats", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("residual", c_void_p), ("r_stats", c_void_p), ("r_gamma", c_void_p), ("r_beta", c_void_p),
                ("out", c_void_p), ("n", c_int), ("reverse", c_int), ("spatial", c_long)]


class TowerDesc(Structure):
    """mvsn_tower_desc"""
    _fields_ = [("inp", c_void_p * 3), ("channels", c_int * 3), ("sample_mod", c_int * 3), ("block_scale", c_void_p),
                ("scale_mod", c_int), ("scale_block", c_int), ("head_chunks", c_int), ("n_blocks", c_int),
                ("dilation", c_int * 6), ("weights", c_void_p), ("params", c_void_p), ("tail_mode", c_int),
                ("prior", c_void_p), ("fx", c_void_p), ("fx_mod", c_int), ("out", c_void_p)]


CONV_FP32, CONV_BF16X3, CONV_FP32_WINO, CONV_BF16 = 0, 1, 2, 3
CHAIN_AUTO, CHAIN_DIRECT, CHAIN_WINOGRAD, CHAIN_STEPWISE, CHAIN_BANDED = 0, 1, 2, 3, 4
ABI_VERSION = 5


# name -> (restype, argtypes); mirrors include/mvsn_hip.h one to one
SIGNATURES = {
    "mvsn_abi_version": (c_int, []),
    "mvsn_last_error": (c_char_p, []),
    "mvsn_plane_sweep_setup": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 5 + [c_void_p]),
    "mvsn_plane_sweep_setup_sources": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p] * 5 + [c_void_p]),
    "mvsn_gather_focal": (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]),
    "mvsn_homography_warp": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p] * 2 + [c_void_p]),
    "mvsn_feature_refiner_packed_floats": (c_size_t, []),
    "mvsn_pack_feature_refiner": (c_int, [c_void_p] * 11 + [c_void_p]),
    "mvsn_incremental_cost_volume_workspace_bytes": (c_size_t, [c_int] * 3),
    "mvsn_incremental_cost_volume_form": (c_int, [c_int] * 2),
    "mvsn_incremental_cost_volume_form_for": (c_int, [c_int] * 3),
    "mvsn_incremental_cost_volume_workspace_bytes_for": (c_size_t, [c_int] * 5),
    "mvsn_incremental_cost_volume_status_offset": (c_size_t, [c_int] * 3),
    "mvsn_incremental_cost_volume_banded_groups": (c_int, [c_int] * 3),
    "mvsn_incremental_cost_volume": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 4 + [c_size_t, c_int, c_void_p]),
    "mvsn_incremental_cost_volume_repair_workspace_bytes": (c_size_t, [c_int] * 3),
    "mvsn_incremental_cost_volume_guarded": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 4 +
                                             [c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mvsn_incremental_cost_volume_bf16": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 4 +
                                          [c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mvsn_conv_bf16x3_supported": (c_int, [POINTER(ConvDesc)]),
    "mvsn_conv_forward_bf16_storage": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int, c_void_p, c_void_p]),
    "mvsn_conv_winograd_supported": (c_int, [POINTER(ConvDesc)]),
    "mvsn_conv_packed_floats": (c_size_t, [POINTER(ConvDesc)]),
    "mvsn_conv_pack_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "mvsn_conv_num_tiles": (c_int, [POINTER(ConvDesc)]),
    "mvsn_conv_forward": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 10 + [c_void_p]),
    "mvsn_conv_forward_blocks": (c_int, [POINTER(ConvDesc), c_void_p, POINTER(c_int), c_int] + [c_void_p] * 4 + [c_void_p]),
    "mvsn_conv_forward_carry": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 8 + [POINTER(ApplyJob), POINTER(c_int), c_void_p]),
    "mvsn_groupnorm_finalize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mvsn_groupnorm_finalize_split_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mvsn_groupnorm_finalize_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvsn_groupnorm_lrelu_apply": (c_int, [c_void_p] * 5 + [c_int, c_long, c_void_p, c_void_p]),
    "mvsn_groupnorm_lrelu_add2": (c_int, [c_void_p] * 8 + [c_int, c_long, c_void_p, c_void_p]),
    "mvsn_conv_to1_block": (c_int, [c_void_p] * 9 + [c_int] * 3 + [c_void_p, c_void_p]),
    "mvsn_debug_set_band_flags": (c_int, [c_int]),
    "mvsn_copy_many": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mvsn_conv_to1_block_records": (c_int, [c_void_p] * 2 + [c_int] + [c_void_p] * 7 + [c_int] * 3 + [c_void_p, c_void_p]),
    "mvsn_groupnorm_lrelu_apply_records": (c_int, [c_void_p] * 2 + [c_int] + [c_void_p] * 6 + [c_int, c_long, c_void_p, c_void_p]),
    "mvsn_conv_to1_supported": (c_int, [c_int, c_int]),
    "mvsn_conv_to1_volume_norm": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p, c_void_p]),
    "mvsn_conv_to1": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_void_p]),
    "mvsn_soft_argmin": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p, c_void_p]),
    "mvsn_channel_l2_norm": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "mvsn_idepth_scale": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "mvsn_refiner_epilogue": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "mvsn_upsample_bilinear": (c_int, [c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "mvsn_upsample_prior": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 3),
    "mvsn_upsample_mask": (c_int, [c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "mvsn_image_pyramid_supported": (c_int, [c_int] * 3),
    "mvsn_image_pyramid": (c_int, [c_void_p] + [c_int] * 5 + [POINTER(c_void_p), c_void_p]),
    "mvsn_prepare_cameras": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p] * 5 + [c_void_p]),
    "mvsn_area_downsample": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_void_p]),
    "mvsn_fuse_sources": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p] * 3 + [c_void_p]),
    "mvsn_idepth_reproject_blocks": (c_int, [c_int]),
    "mvsn_idepth_reproject": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 6 + [c_void_p]),
    "mvsn_occlusion_mask": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p, c_void_p]),
    "mvsn_masked_l1": (c_int, [c_void_p] * 4 + [c_long, c_int, c_void_p, c_void_p]),
    "mvsn_tower_16x32": (c_int, [POINTER(TowerDesc), c_int, c_void_p]),
    "mvsn_depth_metrics_blocks": (c_int, [c_long]),
    "mvsn_depth_metrics": (c_int, [c_void_p] * 3 + [c_int, c_long, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "mvsn_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvsn_gather_strided": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "mvsn_selftest_mfma": (c_int, [c_void_p]),
}


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and type the library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m multi_view_stereonet_amd.build` or __graft_entry__.build()). "
            "There is no CPU fallback for the plane-sweep path.")
    lib = ctypes.CDLL(_LIB_PATH)
    lib.mvsn_abi_version.restype, lib.mvsn_abi_version.argtypes = c_int, []
    if lib.mvsn_abi_version() != ABI_VERSION:     # (checked first: a stale binary lacks the newer symbols)
        raise RuntimeError(f"libmvsn_hip.so is ABI version {lib.mvsn_abi_version()}, this package needs {ABI_VERSION}: "
                           "rebuild it (python -m multi_view_stereonet_amd.build --force)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the binary disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmvsn_hip.so takes device (HBM) pointers; got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("libmvsn_hip.so takes dense tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mvsn_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


# ---- the right to launch CO-RESIDENT workgroups on a device (the banded / slab chain forms) -------------------------
# Those forms spin on sibling workgroups: every workgroup of a launch must be on a CU at the same time, so only ONE
# such launch may be in flight per device (include/mvsn_hip.h).  Inside a process the module orders its forwards; across
# PROCESSES that share a GPU nothing did -- two banded launches could each hold part of the chip and time out into the
# repair launch.  An advisory file lock keyed by the GPU's identity settles it: the first process that wants the banded
# form on a device takes the lock for its lifetime; every other process gets `False` here and AUTO gives it the
# single-launch forms (plane-resident Winograd / stepwise / direct), which need no co-residency and merely queue behind
# the owner's kernels.  (Processes that cannot see each other's lock directory -- separate containers on one GPU -- are
# still covered by the in-stream repair, multi_view_stereonet.py: check_device_status.)
_coresident = {}      # device index -> (granted, open file or None)


def _device_key(index: int) -> str:
    p = torch.cuda.get_device_properties(index)
    uuid = getattr(p, "uuid", None)
    if uuid is not None and str(uuid).strip("0-") != "":
        return str(uuid)
    return "pci-%04x-%02x-%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", index),
                                   getattr(p, "pci_device_id", 0))


def coresident_lock_path(index: int) -> str:
    root = os.environ.get("MVSN_LOCK_DIR") or "/tmp"
    return os.path.join(root, "mvsn_coresident_%s.lock" % _device_key(index).replace("/", "_"))


def coresident_right(index: int) -> bool:
    """True if THIS process may launch the co-resident (banded / slab) chain forms on device `index`: it holds the
    device's advisory lock (taken on first request, kept until the process exits or `release_coresident_right`).
    MVSN_CORESIDENT_LOCK=0 switches the lock off (every process is granted; the in-stream repair is then the only
    guard).  A lock directory that cannot be written grants too -- the pre-lock behaviour, never a silent slowdown."""
    if index in _coresident:
        return _coresident[index][0]
    if os.environ.get("MVSN_CORESIDENT_LOCK", "1") == "0":
        _coresident[index] = (True, None)
        return True
    import fcntl
    try:      # (world-writable: the other process on this GPU may belong to another user)
        old_mask = os.umask(0)
        try:
            f = os.fdopen(os.open(coresident_lock_path(index), os.O_RDWR | os.O_CREAT, 0o666), "r+")
        finally:
            os.umask(old_mask)
    except OSError:
        _coresident[index] = (True, None)
        return True
    try:
        fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
    except OSError:                    # another process owns the device's co-resident launches
        f.close()
        _coresident[index] = (False, None)
        return False
    try:
        f.seek(0)
        f.truncate()
        f.write("%d\n" % os.getpid())
        f.flush()
    except OSError:
        pass
    _coresident[index] = (True, f)
    return True


def release_coresident_right(index=None):
    """Give the lock(s) back (and forget refusals: the next request asks again)."""
    for i in ([index] if index is not None else list(_coresident)):
        granted, f = _coresident.pop(i, (False, None))
        if f is not None:
            f.close()
