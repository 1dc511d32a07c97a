"""Depth-metric and averaging harness.

Mirrors, by name and meaning, the evaluation bookkeeping of the reference's ``test.py``:

* ``get_depth_prediction_metrics``  <- test.py:41-71   (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)
* ``depth_range``                   <- test.py:166-186 (GTA-SfM: (0, 1e3); DeMoN: DPSNet's (0.5, 10))
* ``image_metric_row``              <- test.py:210-235, 258: idepth/baseline -> depth where positive,
                                       mask = truth and estimate both inside the range
* ``compute_avg_metrics``           <- test.py:146-164 (unweighted mean over rows)
* ``evaluate``                      <- test.py:188-280 without the file I/O: one metric row per image,
                                       optionally sharded over ranks and all-gathered (distributed.py)

Pure numpy/torch host code; the forward it drives is the HIP path.
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
    rows: List[List[float]] = []
    idx: List[int] = []
    image_index = 0
    for bi, batch in enumerate(batches):
        bsz = batch["left_image"].shape[0]
        if image_indices is not None or bi % world == rank:
            inputs = snu.multi_view_unpack_batch(batch, device, stereo_network.num_levels)
            outputs = snu.multi_view_forward(stereo_network, inputs, params)
            depth_est = idepth_to_depth(outputs["left_idepthmap_pyr"][0], inputs["baseline"])
            depth_true = inputs["left_depthmap_true"] * inputs["baseline"].view(-1, 1, 1, 1)
            for b in range(bsz):
                row = image_metric_row(depth_true[b, 0].cpu().numpy(), depth_est[b, 0].cpu().numpy(), min_depth,
                                       max_depth)
                if row is not None:
                    rows.append([row[k] for k in METRIC_KEYS] + [outputs["stereo_time_ms"] / bsz])
                    idx.append(int(image_indices[image_index + b]) if image_indices is not None else image_index + b)
        image_index += bsz
    dev = device if (world > 1 and torch.distributed.get_backend() == "nccl") else torch.device("cpu")
    rt = torch.tensor(rows, dtype=torch.float64, device=dev).reshape(-1, len(METRIC_KEYS) + 1)
    it = torch.tensor(idx, dtype=torch.int64, device=dev)
    all_rows, _ = mdist.gather_metric_rows(rt, it)
    avg = mdist.average_rows(all_rows).tolist()
    out = {k: avg[i] for i, k in enumerate(METRIC_KEYS)} if all_rows.shape[0] else {}
    if all_rows.shape[0]:
        out["runtime_ms"] = avg[-1]
    out["num_samples"] = int(all_rows.shape[0])
    return out
