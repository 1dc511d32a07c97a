"""Host-side callers of the plane-sweep forward.

Mirrors, by name and argument meaning, the three functions of the reference that sit
directly either side of ``MultiViewStereoNet.forward``:

* ``build_image_pyramid``      <- utils/image_utils.py:111-128
* ``multi_view_unpack_batch``  <- multi_view_stereonet/multi_view_stereonet_utils.py:541-641
* ``multi_view_forward``       <- multi_view_stereonet/multi_view_stereonet_utils.py:643-662
* ``unpack_batch`` / ``forward`` (two-view twins, optional right-view estimate) <- :406-539

Nothing here is compute-heavy; it is tensor plumbing on whatever device it is handed
(PyTorch-ROCm on the GPU box, CPU in the oracle tests).
"""
import time
from typing import Dict, List

import torch
import torch.nn.functional as F


def build_image_pyramid(image: torch.Tensor, num_levels: int) -> List[torch.Tensor]:
    """Ceil-halving area pyramid: level l has ((h+1)//2, (w+1)//2) of level l-1.

    ``interpolate(mode="area")`` is an adaptive average pool, i.e. an exact 2x2 mean for
    even sizes and overlapping windows for odd ones (utils/image_utils.py:118-126).
    """
    if image.dim() != 4:
        raise AssertionError("image must be (batch, channels, rows, cols)")
    if image.is_cuda and num_levels >= 2:
        # device frames whose sizes halve exactly: every level in ONE pass over the frames (mvsn_image_pyramid)
        import ctypes
        from . import _native
        lib = _native.load()
        n, c, rows, cols = image.shape
        if lib.mvsn_image_pyramid_supported(rows, cols, num_levels):
            src = image.contiguous().float()
            outs = [torch.empty((n, c, rows >> l, cols >> l), dtype=torch.float32, device=image.device)
                    for l in range(1, num_levels)]
            ptrs = (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            _native.check(lib.mvsn_image_pyramid(_native.ptr(src), n, c, rows, cols, num_levels, ptrs,
                                                 _native.stream()), "mvsn_image_pyramid")
            return [image] + outs
    levels = [image]
    while len(levels) < num_levels:
        prev = levels[-1]
        size = ((prev.shape[2] + 1) // 2, (prev.shape[3] + 1) // 2)
        if prev.is_cuda:
            # device frames: the HIP pyramid kernel (no ATen fallback; raises if the library is missing)
            from . import _native
            lib = _native.load()
            prev = prev.contiguous().float()
            nxt = torch.empty(prev.shape[:2] + size, dtype=torch.float32, device=prev.device)
            _native.check(lib.mvsn_area_downsample(_native.ptr(prev), prev.shape[0], prev.shape[1], prev.shape[2],
                                                   prev.shape[3], _native.ptr(nxt), _native.stream()),
                          "mvsn_area_downsample")
            levels.append(nxt)
        else:
            levels.append(F.adaptive_avg_pool2d(prev, size))   # host-side input prep for the oracle/tests
    return levels


def build_intrinsics_pyramid(K: torch.Tensor, image_pyr: List[torch.Tensor]) -> List[torch.Tensor]:
    """Per-level intrinsics with the half-pixel-aware principal point.

    A resize by s maps pixel centre x to s*(x+0.5)-0.5, hence cx' = s*(cx+0.5)-0.5 and
    fx' = s*fx, with s taken from the actual level sizes as python floats
    (multi_view_stereonet_utils.py:575-581).
    """
    rows0, cols0 = image_pyr[0].shape[-2], image_pyr[0].shape[-1]
    out = [K]
    for lvl in range(1, len(image_pyr)):
        sx = float(image_pyr[lvl].shape[-1]) / cols0
        sy = float(image_pyr[lvl].shape[-2]) / rows0
        Kl = K.clone()
        Kl[:, 0, 0] *= sx
        Kl[:, 1, 1] *= sy
        Kl[:, 0, 2] = sx * (Kl[:, 0, 2] + 0.5) - 0.5
        Kl[:, 1, 2] = sy * (Kl[:, 1, 2] + 0.5) - 0.5
        out.append(Kl)
    return out


def _prepare_cameras_on_device(K: torch.Tensor, poses, image_pyr: List[torch.Tensor], device):
    from . import _native
    lib = _native.load()
    K = K.float().contiguous()
    B, S, L = K.shape[0], len(poses), len(image_pyr)
    T = torch.stack([p.to(device).squeeze(1).float() for p in poses], 0).contiguous()          # (S,B,4,4)
    sizes = torch.tensor([v for lvl in image_pyr for v in lvl.shape[-2:]], dtype=torch.int32, device=device)
    K_pyr = torch.empty((L, B, 4, 4), dtype=torch.float32, device=device)
    Tn, Ti = torch.empty_like(T), torch.empty_like(T)
    baseline = torch.empty((B,), dtype=torch.float32, device=device)
    _native.check(lib.mvsn_prepare_cameras(_native.ptr(K), _native.ptr(T), B, S, L, _native.ptr(sizes), _native.ptr(K_pyr),
                                           _native.ptr(Tn), _native.ptr(Ti), _native.ptr(baseline), _native.stream()),
                  "mvsn_prepare_cameras")
    return [K_pyr[l] for l in range(L)], [Tn[s] for s in range(S)], [Ti[s] for s in range(S)], baseline


class Prefetcher:
    """Overlapped input feed: iterates DataLoader-style batch dicts and hands them over WITH their tensors already
    on `device`, the host-to-device copies of batch k+1 travelling on a side stream while batch k computes (two
    batches in flight; tensors that are not pinned yet are pinned first).  What the reference does serially in
    multi_view_unpack_batch (``.to(device)`` per tensor, multi_view_stereonet_utils.py:545-549, 553, 588) --
    measured on MI355X at the headline config: 1,938 depthmaps/s with blocking copies, 3,278/s overlapped.
    On a CPU device (tests) it is a plain pass-through.

        for batch in Prefetcher(loader, device):
            inputs = multi_view_unpack_batch(batch, device, 5)     # tensors already resident: no copy, no sync
    """

    def __init__(self, batches, device, pin: bool = True):
        self.batches, self.device, self.pin = batches, torch.device(device), pin
        self.on_gpu = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.on_gpu else None

    def _move(self, v):
        if torch.is_tensor(v):
            if self.pin and not v.is_cuda and not v.is_pinned():
                v = v.pin_memory()
            return v.to(self.device, non_blocking=True)
        if isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0]):
            return [self._move(x) for x in v]
        return v

    def _stage(self, batch):
        with torch.cuda.stream(self.side):
            moved = {k: self._move(v) for k, v in batch.items()}
            ev = torch.cuda.Event()
            ev.record(self.side)
        return moved, ev

    def __iter__(self):
        if not self.on_gpu:
            yield from self.batches
            return
        it = iter(self.batches)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))       # the next batch's copies are enqueued before this one is used
            except StopIteration:
                nxt = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            for v in cur.values():
                for t in (v if isinstance(v, list) else [v]):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)
            yield cur


def multi_view_unpack_batch(batch: Dict[str, object], device, num_levels: int,
                            check_baseline: bool = True) -> Dict[str, object]:
    """DataLoader batch -> forward() inputs.

    ``check_baseline=False`` defers the reference's "baseline must be positive" assertion (a device-to-host sync per
    batch) to the caller: ``inputs["baseline_ok"]`` is then a device-side boolean to be looked at once, later.

    Image pyramids for the reference view and every source view, the K pyramid, the
    source poses and their inverses with ALL translations divided by the baseline to the
    FIRST source (multi_view_stereonet_utils.py:597-604), and that baseline.  When the batch
    carries ground-truth depth it is expressed in the same baseline units and inverted
    where positive (:615-637).
    """
    left = batch["left_image"].to(device)
    rights = [r.to(device) for r in batch["right_image"]]

    left_pyr = build_image_pyramid(left, num_levels)
    K = batch["K"].to(device).squeeze(1)
    right_pyrs = [build_image_pyramid(r, num_levels) for r in rights]

    if K.is_cuda:
        # cameras in one launch: K pyramid, poses and inverses normalised by the first source's baseline
        K_pyr, T_r_in_l, T_l_in_r, baseline = _prepare_cameras_on_device(K, batch["T_right_in_left"], left_pyr, device)
    else:
        K_pyr = build_intrinsics_pyramid(K, left_pyr)
        T_r_in_l, T_l_in_r = [], []
        for pose in batch["T_right_in_left"]:
            T = pose.to(device).squeeze(1).clone()
            T_r_in_l.append(T)
            T_l_in_r.append(torch.linalg.inv(T))
        baseline = T_r_in_l[0][:, :3, 3].pow(2).sum(1).sqrt()
        for T, Tinv in zip(T_r_in_l, T_l_in_r):
            T[:, :3, 3] /= baseline[:, None]
            Tinv[:, :3, 3] /= baseline[:, None]
    baseline_ok = (baseline > 0).all()
    if check_baseline and not bool(baseline_ok):
        raise AssertionError("baseline to the first source view must be positive")

    inputs = {"left_filename": batch.get("left_filename"),
              "baseline_ok": baseline_ok,
              "right_filename": batch.get("right_filename"),
              "T_right_in_left": T_r_in_l,
              "T_left_in_right": T_l_in_r,
              "K_pyr": K_pyr,
              "left_image_pyr": left_pyr,
              "right_image_pyr": right_pyrs,
              "baseline": baseline}

    if "left_depthmap_true" in batch:
        scale = baseline.view(-1, 1, 1, 1)
        depth = batch["left_depthmap_true"].to(device) / scale
        inputs["left_depthmap_true"] = depth
        inputs["left_idepthmap_true"] = torch.where(depth > 0, 1.0 / depth, depth)
        rd = [d.to(device) / scale for d in batch["right_depthmap_true"]]
        inputs["right_depthmap_true"] = rd
        inputs["right_idepthmap_true"] = [torch.where(d > 0, 1.0 / d, d) for d in rd]

    if inputs["left_image_pyr"][0].dtype != torch.float32:
        raise AssertionError("images must be float32")
    return inputs


def unpack_batch(batch: Dict[str, object], device, num_levels: int) -> Dict[str, object]:
    """Two-view twin of multi_view_unpack_batch (multi_view_stereonet_utils.py:406-501): one source
    view, tensors instead of lists, the pose normalised by its own baseline."""
    left = batch["left_image"].to(device)
    right = batch["right_image"].to(device)
    K = batch["K"].to(device).squeeze(1)
    T = batch["T_right_in_left"].to(device).squeeze(1).clone()
    baseline = T[:, :3, 3].pow(2).sum(1).sqrt()
    if not bool((baseline > 0).all()):
        raise AssertionError("baseline must be positive")
    T[:, :3, 3] /= baseline[:, None]
    left_pyr = build_image_pyramid(left, num_levels)
    inputs = {"left_filename": batch.get("left_filename"), "right_filename": batch.get("right_filename"),
              "T_right_in_left": T, "T_left_in_right": torch.linalg.inv(T),
              "K_pyr": build_intrinsics_pyramid(K, left_pyr), "left_image_pyr": left_pyr,
              "right_image_pyr": build_image_pyramid(right, num_levels), "baseline": baseline}
    for key in ("left_disparity_true", "right_disparity_true"):
        if key in batch:
            inputs[key] = batch[key].to(device)
    if "left_depthmap_true" in batch:
        scale = baseline.view(-1, 1, 1, 1)
        for side in ("left", "right"):
            depth = batch[f"{side}_depthmap_true"].to(device) / scale
            inputs[f"{side}_depthmap_true"] = depth
            inputs[f"{side}_idepthmap_true"] = torch.where(depth > 0, 1.0 / depth, depth)
    if inputs["left_image_pyr"][0].dtype != torch.float32:
        raise AssertionError("images must be float32")
    return inputs


def forward(stereo_network, inputs: Dict[str, object], params: Dict[str, object]):
    """Two-view twin of multi_view_forward (multi_view_stereonet_utils.py:503-539).  With
    ``params["estimate_right_idepthmap"]`` the network runs a second time with the views swapped
    (the source becomes the reference, the pose is inverted) and the two times are averaged."""
    on_gpu = inputs["left_image_pyr"][0].is_cuda
    D = int(params["num_idepth_samples"])
    flt = bool(params.get("cost_volume_filter", True))
    refs = list(params.get("refiners", [True] * 5))

    def run(ref_pyr, pose, src_pyr):
        a, b = _tick(on_gpu)
        out = stereo_network(ref_pyr, inputs["K_pyr"], [pose], [src_pyr], D, flt, refs)
        return out, _tock(on_gpu, a, b)

    left, ms = run(inputs["left_image_pyr"], inputs["T_right_in_left"], inputs["right_image_pyr"])
    outputs = {"left_idepthmap_pyr": left["left_idepthmap_pyr"],
               "left_idepthmap_raw_pyr": left["left_idepthmap_raw_pyr"],
               "left_idepthmap_mask_pyr": left["left_idepthmap_mask_pyr"], "stereo_time_ms": ms}
    if params.get("estimate_right_idepthmap", False):
        right, ms_r = run(inputs["right_image_pyr"], inputs["T_left_in_right"], inputs["left_image_pyr"])
        outputs["right_idepthmap_pyr"] = right["left_idepthmap_pyr"]
        outputs["right_idepthmap_raw_pyr"] = right["left_idepthmap_raw_pyr"]
        outputs["right_idepthmap_mask_pyr"] = right["left_idepthmap_mask_pyr"]
        outputs["stereo_time_ms"] = 0.5 * (ms + ms_r)
    return outputs


def occlusion_masks(inputs: Dict[str, object], outputs: Dict[str, object]) -> Dict[str, list]:
    """The occlusion-mask pyramids the reference derives from a bidirectional estimate
    (multi_view_stereonet_utils.py:711-730): per level, where a left pixel is hidden in the right view and vice
    versa.  `outputs` must come from forward(..., estimate_right_idepthmap=True).  Levels without an estimate stay
    None.  Both pyramids feed losses.left_right_idepthmap_consistency_losses (:749-753)."""
    from . import losses
    n = len(outputs["left_idepthmap_pyr"])
    left_occ, right_occ = [None] * n, [None] * n
    for lvl in range(n):
        if outputs["left_idepthmap_pyr"][lvl] is None:
            continue
        left_occ[lvl] = losses.get_occlusion_mask(
            inputs["K_pyr"][lvl], inputs["T_right_in_left"], outputs["left_idepthmap_pyr"][lvl],
            outputs["left_idepthmap_mask_pyr"][lvl], outputs["right_idepthmap_pyr"][lvl],
            outputs["right_idepthmap_mask_pyr"][lvl])
        right_occ[lvl] = losses.get_occlusion_mask(
            inputs["K_pyr"][lvl], inputs["T_left_in_right"], outputs["right_idepthmap_pyr"][lvl],
            outputs["right_idepthmap_mask_pyr"][lvl], outputs["left_idepthmap_pyr"][lvl],
            outputs["left_idepthmap_mask_pyr"][lvl])
    return {"left_occlusion_mask_pyr": left_occ, "right_occlusion_mask_pyr": right_occ}


def _tick(device_is_gpu: bool):
    if device_is_gpu:
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        return a, b
    return time.time(), None


def _tock(device_is_gpu: bool, a, b) -> float:
    if device_is_gpu:
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)
    return (time.time() - a) * 1000.0


def multi_view_forward(stereo_network, inputs: Dict[str, object], params: Dict[str, object],
                       sync_timer: bool = True):
    """Time and run the network exactly as the reference's wrapper does.

    Timer semantics follow utils/pytorch_utils.py:31-48 (device events bracketed by
    synchronize on a GPU, wall clock on CPU).  ``cost_volume_filter`` / ``refiners``
    default to on when the yaml lacks them (the DeMoN params.yaml does).
    ``sync_timer=False`` (throughput loops): the two device events are recorded without any synchronize and returned
    as ``stereo_time_events``; ``stereo_time_ms`` is None until the caller reads ``a.elapsed_time(b)`` after its own sync.
    """
    on_gpu = inputs["left_image_pyr"][0].is_cuda
    if on_gpu and not sync_timer:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = stereo_network(inputs["left_image_pyr"], inputs["K_pyr"], inputs["T_right_in_left"],
                             inputs["right_image_pyr"], int(params["num_idepth_samples"]),
                             bool(params.get("cost_volume_filter", True)),
                             list(params.get("refiners", [True] * 5)))
        b.record()
        return {"left_idepthmap_pyr": out["left_idepthmap_pyr"],
                "left_idepthmap_raw_pyr": out["left_idepthmap_raw_pyr"],
                "left_idepthmap_mask_pyr": out["left_idepthmap_mask_pyr"],
                "stereo_time_ms": None, "stereo_time_events": (a, b)}
    a, b = _tick(on_gpu)
    out = stereo_network(inputs["left_image_pyr"], inputs["K_pyr"], inputs["T_right_in_left"],
                         inputs["right_image_pyr"], int(params["num_idepth_samples"]),
                         bool(params.get("cost_volume_filter", True)),
                         list(params.get("refiners", [True] * 5)))
    ms = _tock(on_gpu, a, b)
    if on_gpu and hasattr(stereo_network, "check_device_status"):
        # (this path only: _tock has just synchronised.  The sync_timer=False path above returns without a status check --
        # its caller synchronises once at the end and checks then, metrics.evaluate)
        stereo_network.check_device_status(synchronize=False)
    return {"left_idepthmap_pyr": out["left_idepthmap_pyr"],
            "left_idepthmap_raw_pyr": out["left_idepthmap_raw_pyr"],
            "left_idepthmap_mask_pyr": out["left_idepthmap_mask_pyr"],
            "stereo_time_ms": ms}
