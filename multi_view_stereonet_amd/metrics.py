"""Depth-metric and averaging harness.

Mirrors, by name and meaning, the evaluation bookkeeping of the reference's ``test.py``:

* ``get_depth_prediction_metrics``  <- test.py:41-71   (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)
* ``depth_range``                   <- test.py:166-186 (GTA-SfM: (0, 1e3); DeMoN: DPSNet's (0.5, 10))
* ``image_metric_row``              <- test.py:210-235, 258: idepth/baseline -> depth where positive,
                                       mask = truth and estimate both inside the range
* ``compute_avg_metrics``           <- test.py:146-164 (unweighted mean over rows)
* ``evaluate``                      <- test.py:188-280 without the file I/O: one metric row per image,
                                       optionally sharded over ranks and all-gathered (distributed.py)

``evaluate`` keeps everything on the device when the network is on a GPU: inputs arrive through
``multi_view_stereonet_utils.Prefetcher`` (copies of batch k+1 under batch k's forward), the metric sums of a batch are
one HBM pass (``depth_metric_rows`` -> mvsn_depth_metrics: nine doubles per image), nothing is copied back or
synchronised per image, and the rows are all-gathered once at the end.  The numpy functions are the host form of the
same arithmetic (CPU devices, and what the device rows are tested against).
"""
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as mdist
from . import multi_view_stereonet_utils as snu

METRIC_KEYS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def get_depth_prediction_metrics(depthmap_true: np.ndarray, depthmap_est: np.ndarray) -> Dict[str, float]:
    """KITTI-style metrics over already-masked, strictly positive depths."""
    t = np.asarray(depthmap_true)
    e = np.asarray(depthmap_est)
    ratio = np.maximum(t / e, e / t)
    diff = t - e
    return {"abs_rel": np.mean(np.abs(diff) / t),
            "sq_rel": np.mean(diff ** 2 / t),
            "rmse": np.sqrt(np.mean(diff ** 2)),
            "rmse_log": np.sqrt(np.mean((np.log(t) - np.log(e)) ** 2)),
            "a1": np.mean(ratio < 1.25),
            "a2": np.mean(ratio < 1.25 ** 2),
            "a3": np.mean(ratio < 1.25 ** 3)}


def depth_range(split: str) -> Tuple[float, float]:
    if "gta_sfm" in split:
        return 0.0, 1e3
    if "demon" in split:
        return 0.5, 10.0
    raise ValueError(f"unknown split {split!r}")


def idepth_to_depth(idepth: torch.Tensor, baseline: torch.Tensor) -> torch.Tensor:
    """Network idepth (unit-baseline) -> metric depth; non-positive idepths stay as they are."""
    scaled = idepth / baseline.view(-1, 1, 1, 1)
    return torch.where(scaled > 0, 1.0 / scaled, scaled)


def image_metric_row(depth_true: np.ndarray, depth_est: np.ndarray, min_depth: float,
                     max_depth: float) -> Optional[Dict[str, float]]:
    """None when the image has no valid ground truth (the reference skips it, test.py:223-225)."""
    mask = (depth_true > min_depth) & (depth_true < max_depth)
    if mask.sum() <= 0:
        return None
    mask = mask & (depth_est > min_depth) & (depth_est < max_depth)
    return get_depth_prediction_metrics(depth_true[mask], depth_est[mask])


ROW_KEYS = ("n_truth", "n_selected") + METRIC_KEYS


def depth_metric_rows(idepth_est: torch.Tensor, depth_true: torch.Tensor, baseline: torch.Tensor, min_depth: float,
                      max_depth: float) -> torch.Tensor:
    """(B, 9) float64 rows {n_truth, n_selected, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3} for a batch, computed
    where the tensors live.  On a HIP device: mvsn_depth_metrics (idepth -> depth, both range masks, the seven means;
    test.py:41-71, 210-235) -- no host copy, no synchronisation.  On CPU: the numpy functions above, image by image."""
    B = idepth_est.shape[0]
    if idepth_est.is_cuda:
        from . import _native
        lib = _native.load()
        est = idepth_est.reshape(B, -1).contiguous().float()
        tru = depth_true.reshape(B, -1).contiguous().float()
        if tru.shape != est.shape:
            raise ValueError("estimate and ground truth must have the same size (test.py:227 assumes it)")
        pixels = est.shape[1]
        base = baseline.reshape(B).contiguous().float()
        blocks = lib.mvsn_depth_metrics_blocks(pixels)
        partials = torch.empty((B, blocks, 9), dtype=torch.float64, device=est.device)
        rows = torch.empty((B, 9), dtype=torch.float64, device=est.device)
        _native.check(lib.mvsn_depth_metrics(_native.ptr(est), _native.ptr(tru), _native.ptr(base), B, pixels,
                                             float(min_depth), float(max_depth), _native.ptr(partials),
                                             _native.ptr(rows), _native.stream()), "mvsn_depth_metrics")
        return rows
    depth_est = idepth_to_depth(idepth_est, baseline)
    rows = torch.full((B, 9), float("nan"), dtype=torch.float64)
    for b in range(B):
        t, e = depth_true[b].reshape(-1).numpy(), depth_est[b].reshape(-1).numpy()
        truth = (t > min_depth) & (t < max_depth)
        sel = truth & (e > min_depth) & (e < max_depth)
        rows[b, 0], rows[b, 1] = float(truth.sum()), float(sel.sum())
        if sel.any():
            m = get_depth_prediction_metrics(t[sel], e[sel])
            rows[b, 2:] = torch.tensor([float(m[k]) for k in METRIC_KEYS], dtype=torch.float64)
    return rows


def compute_avg_metrics(rows: Sequence[Dict[str, float]]) -> Dict[str, float]:
    if not rows:
        return {"num_samples": 0}
    keys = list(rows[0].keys())
    mat = np.array([[r[k] for k in keys] for r in rows], dtype=np.float64)
    out = {k: float(v) for k, v in zip(keys, mat.mean(axis=0))}
    out["num_samples"] = mat.shape[0]
    return out


def evaluate(stereo_network, batches: Iterable[dict], params: dict, split: str, device,
             rank: int = 0, world: int = 1, image_indices: Optional[Sequence[int]] = None) -> Dict[str, float]:
    """Run the network over ``batches`` (DataLoader-style dicts carrying ``left_depthmap_true`` in
    metric units), one metric row per image, averaged over all ranks' rows.

    Sharding, two forms:
      * ``image_indices`` given: ``batches`` already holds only THIS rank's images (the dataset was sharded, e.g.
        ``Subset(data, distributed.shard_indices(len(data), rank, world))``), and ``image_indices[k]`` is the global
        index of the k-th image it yields -- each rank reads and decodes only its own files (evaluate.py does this);
      * otherwise every rank iterates the same ``batches`` and processes batch i when i % world == rank (fine for
        in-memory batches; with a file-backed loader it makes every rank read everything).
    Rows (+ per-image runtime = batch time / batch size) are all-gathered at the end.
    """
    min_depth, max_depth = depth_range(split)
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    row_parts, idx_parts, timers, checks = [], [], [], []
    image_index = 0
    mine = (lambda bi: True) if image_indices is not None else (lambda bi: bi % world == rank)

    def own_batches():
        # (only this rank's batches reach the prefetcher: nothing is copied for the others)
        nonlocal image_index
        for bi, batch in enumerate(batches):
            bsz = batch["left_image"].shape[0]
            if mine(bi):
                yield dict(batch, _first_image=image_index)
            image_index += bsz

    for batch in snu.Prefetcher(own_batches(), device):
        first = batch.pop("_first_image")
        bsz = batch["left_image"].shape[0]
        inputs = snu.multi_view_unpack_batch(batch, device, stereo_network.num_levels, check_baseline=not on_gpu)
        outputs = snu.multi_view_forward(stereo_network, inputs, params, sync_timer=not on_gpu)
        # ground truth in metric units, as the reference reloads it (test.py:216-219); the estimate is converted
        # inside the metric pass
        truth = batch["left_depthmap_true"].to(device)
        rows_b = depth_metric_rows(outputs["left_idepthmap_pyr"][0], truth, inputs["baseline"], min_depth, max_depth)
        row_parts.append(rows_b)
        gidx = torch.arange(first, first + bsz, dtype=torch.int64)
        if image_indices is not None:
            gidx = torch.as_tensor([int(image_indices[i]) for i in range(first, first + bsz)], dtype=torch.int64)
        idx_parts.append(gidx)
        timers.append((outputs.get("stereo_time_events"), outputs["stereo_time_ms"], bsz))
        checks.append(inputs["baseline_ok"])
    # one synchronisation for the whole loop: runtimes, the deferred baseline assertion, the rows
    if on_gpu:
        torch.cuda.synchronize(device)
    if checks and not bool(torch.stack([c.reshape(()) for c in checks]).all()):
        raise AssertionError("baseline to the first source view must be positive")
    if on_gpu and hasattr(stereo_network, "check_device_status"):
        stereo_network.check_device_status(synchronize=False)
    rows_all = torch.cat(row_parts, 0) if row_parts else torch.zeros((0, 9), dtype=torch.float64, device=device)
    idx_all = torch.cat(idx_parts, 0) if idx_parts else torch.zeros((0,), dtype=torch.int64)
    # per-image runtime = batch time / batch size (the reference writes the BATCH time per file, test.py:271; both
    # are reported: `runtime_ms` per image, `batch_runtime_ms` as the reference's column)
    per_image, per_batch = [], []
    for ev, ms, bsz in timers:
        ms = ev[0].elapsed_time(ev[1]) if ev is not None else ms
        per_image += [ms / bsz] * bsz
        per_batch += [ms] * bsz
    dev = device if (on_gpu and (world == 1 or torch.distributed.get_backend() == "nccl")) else torch.device("cpu")
    rows_all = rows_all.to(dev)
    extra = torch.tensor([per_image, per_batch], dtype=torch.float64, device=dev).t().reshape(-1, 2)
    keep = rows_all[:, 0] > 0            # images without valid ground truth are skipped (test.py:223-225)
    rt = torch.cat([rows_all[:, 2:], extra], 1)[keep]
    it = idx_all.to(dev)[keep]
    all_rows, _ = mdist.gather_metric_rows(rt, it)
    avg = mdist.average_rows(all_rows).tolist()
    out = {k: avg[i] for i, k in enumerate(METRIC_KEYS)} if all_rows.shape[0] else {}
    if all_rows.shape[0]:
        out["runtime_ms"] = avg[-2]
        out["batch_runtime_ms"] = avg[-1]
    out["num_samples"] = int(all_rows.shape[0])
    return out
