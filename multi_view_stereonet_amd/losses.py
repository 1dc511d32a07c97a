"""Two-view consistency ops on the HIP path.

Same names, arguments and return values as the reference's ``multi_view_stereonet/losses.py``:

* ``get_occlusion_mask``                         <- losses.py:42-82
* ``left_right_idepthmap_consistency_losses``    <- losses.py:112-160

They consume what the bidirectional forward (``multi_view_stereonet_utils.forward`` with
``estimate_right_idepthmap``) returns.  Inference-side only: no autograd (the reference trains through them; this
build has no backward).  Everything runs in libmvsn_hip.so; CPU tensors raise.
"""
from typing import List, Optional

import torch

from . import _native


def _as_flag_bytes(mask: torch.Tensor) -> torch.Tensor:
    """A mask of any dtype as the dense 0/1 bytes the kernels read (the reference calls .float() on its masks and so
    accepts any dtype, losses.py:129-135; a float / int32 / int64 tensor handed over as raw bytes would be misread)."""
    if mask.dtype not in (torch.bool, torch.uint8):
        mask = mask != 0
    elif mask.dtype == torch.uint8:
        mask = mask != 0            # any non-zero byte counts as set, as .float() > 0 would
    return mask.contiguous()


def _reproject(K, T_other_in_this, idepth, other_idepth, other_mask=None, want_partials=False, want_uv=False):
    if not idepth.is_cuda:
        raise RuntimeError("the consistency ops run in libmvsn_hip.so on HIP devices only; got a CPU tensor")
    lib = _native.load()
    B, _, rows, cols = idepth.shape
    dev = idepth.device
    K, T = K.float().contiguous(), T_other_in_this.float().contiguous()
    idepth, other_idepth = idepth.float().contiguous(), other_idepth.float().contiguous()
    f = dict(dtype=torch.float32, device=dev)
    id_prime, sampled = torch.empty((B, 1, rows, cols), **f), torch.empty((B, 1, rows, cols), **f)
    invalid = torch.empty((B, 1, rows, cols), dtype=torch.bool, device=dev)
    mask_sampled = torch.empty((B, 1, rows, cols), dtype=torch.bool, device=dev) if other_mask is not None else None
    om = _as_flag_bytes(other_mask) if other_mask is not None else None
    uv = torch.empty((B, rows, cols, 2), **f) if want_uv else None
    partials = torch.empty((B, lib.mvsn_idepth_reproject_blocks(rows * cols)), **f) if want_partials else None
    _native.check(lib.mvsn_idepth_reproject(_native.ptr(K), _native.ptr(T), _native.ptr(idepth), _native.ptr(other_idepth),
                                            _native.ptr(om), B, rows, cols, _native.ptr(id_prime), _native.ptr(sampled),
                                            _native.ptr(mask_sampled), _native.ptr(invalid), _native.ptr(uv),
                                            _native.ptr(partials), _native.stream()), "mvsn_idepth_reproject")
    return id_prime, sampled, mask_sampled, invalid, uv, partials


def idepthmap_projector(K, T_right_in_left, left_idepthmap):
    """IDepthmapProjector.forward (stereo/image_predictor.py:538-576): (right_pixels, right_idepths, mask)."""
    id_prime, _, _, invalid, uv, _ = _reproject(K, T_right_in_left, left_idepthmap, left_idepthmap, want_uv=True)
    return uv, id_prime, invalid


def get_occlusion_mask(K, T_right_in_left, left_idepthmap, left_invalid_mask, right_idepthmap, right_invalid_mask):
    """Left mask that is 1 where a pixel is occluded in the right view (the two *_invalid_mask arguments are accepted
    and unused, exactly as in the reference, losses.py:76-78)."""
    lib = _native.load()
    B, _, rows, cols = left_idepthmap.shape
    id_prime, sampled, _, invalid, _, partials = _reproject(K, T_right_in_left, left_idepthmap, right_idepthmap,
                                                            want_partials=True)
    mask = torch.empty((B, 1, rows, cols), dtype=torch.bool, device=left_idepthmap.device)
    _native.check(lib.mvsn_occlusion_mask(_native.ptr(id_prime), _native.ptr(sampled), _native.ptr(invalid),
                                          _native.ptr(partials), B, rows * cols, _native.ptr(mask), _native.stream()),
                  "mvsn_occlusion_mask")
    return mask


def left_right_idepthmap_consistency_losses(T_right_in_left, T_left_in_right, K_pyr, left_idepthmap_pyr: List[Optional[torch.Tensor]],
                                            left_occlusion_mask_pyr, right_idepthmap_pyr, right_occlusion_mask_pyr):
    """Left/right geometric consistency between idepth pyramids: 0-dim tensor on the inputs' device."""
    lib = _native.load()
    loss = None
    for lvl in range(len(left_idepthmap_pyr)):
        if left_idepthmap_pyr[lvl] is None:
            continue
        for T, a, a_occ, b, b_occ in ((T_right_in_left, left_idepthmap_pyr[lvl], left_occlusion_mask_pyr[lvl],
                                       right_idepthmap_pyr[lvl], right_occlusion_mask_pyr[lvl]),
                                      (T_left_in_right, right_idepthmap_pyr[lvl], right_occlusion_mask_pyr[lvl],
                                       left_idepthmap_pyr[lvl], left_occlusion_mask_pyr[lvl])):
            projected, sampled, occ_sampled, _, _, _ = _reproject(K_pyr[lvl], T, a, b, other_mask=b_occ)
            first = loss is None
            if first:
                loss = torch.empty((1,), dtype=torch.float32, device=a.device)
            a_occ = _as_flag_bytes(a_occ)
            _native.check(lib.mvsn_masked_l1(_native.ptr(projected), _native.ptr(sampled), _native.ptr(a_occ),
                                             _native.ptr(occ_sampled), projected.numel(), 0 if first else 1,
                                             _native.ptr(loss), _native.stream()), "mvsn_masked_l1")
    if loss is None:
        return torch.zeros(())
    return loss[0]
