import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def unpack_mask(fix, lvl):
    shape = tuple(int(x) for x in fix[f"mask_shape_{lvl}"])
    n = int(np.prod(shape))
    return np.unpackbits(fix[f"mask_{lvl}"])[:n].reshape(shape).astype(bool)


def batch_from_meta(meta, jitter=0.0, smooth=False):
    """Regenerate the synthetic batch a fixture was made from."""
    from multi_view_stereonet_amd import synthetic
    rows, cols, D, S, B, seed = (int(x) for x in meta)
    return synthetic.make_batch(rows, cols, S, batch=B, seed=seed, pose_jitter=float(jitter), smooth=bool(smooth)), D


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = b.abs().mean().clamp_min(1e-12)
    return float((a - b).abs().mean() / denom), float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_err_per_pixel(a, ref):
    """The contract's own reading of "within 1e-3 relative on the final depthmap": per-pixel |a - ref| / max(|ref|, floor)
    with floor = 1e-3 * mean|ref|, over the pixels where ref > 0 (the refiner's final relu clamps the others to exactly
    0, multi_view_stereonet.py:482).  Returns (max, p99.9)."""
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    ref = torch.as_tensor(ref, dtype=torch.float64).reshape(-1)
    floor = 1e-3 * ref.abs().mean().clamp_min(1e-12)
    sel = ref > 0
    if not bool(sel.any()):
        return 0.0, 0.0
    e = ((a - ref).abs() / ref.abs().clamp_min(floor))[sel]
    return float(e.max()), float(torch.quantile(e, 0.999)) if e.numel() <= (1 << 24) else float(e.sort().values[int(0.999 * (e.numel() - 1))])


def free_port():
    """A loopback TCP port for a torch.distributed rendezvous (127.0.0.1: the container hostname may not resolve)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p
