"""Seeded synthetic plane-sweep inputs (SURVEY.md section 8d recipe).

There are no datasets in this environment, so every parity test, the golden fixtures and
``bench.py`` use the same generator: uniform-random frames in [-1, 1] (the range the
reference's dataset transform produces, datasets/multi_view_stereo_dataset.py:100-106), a
pin-hole K with focal 0.8*cols and the principal point at the image centre, and source
cameras rotated about y and translated sideways.  A ``smooth`` option renders a band-limited
texture instead of white noise so that cost volumes have real minima.

The output is a *batch dict* in the layout the reference's DataLoader yields and
``multi_view_unpack_batch`` consumes (multi_view_stereonet_utils.py:541-594):
  left_image (B,3,H,W), right_image [S x (B,3,H,W)], K (B,1,4,4), T_right_in_left [S x (B,1,4,4)].
Everything is generated with a CPU ``torch.Generator`` so the numbers are identical on the
build container and on the GPU box.
"""
import math
from typing import Dict, List

import torch


def _smooth_image(gen: torch.Generator, batch: int, rows: int, cols: int) -> torch.Tensor:
    """Sum of a few random low-frequency sinusoids per channel, scaled into [-1, 1]."""
    yy = torch.arange(rows, dtype=torch.float32).view(1, 1, rows, 1) / rows
    xx = torch.arange(cols, dtype=torch.float32).view(1, 1, 1, cols) / cols
    img = torch.zeros(batch, 3, rows, cols)
    for _ in range(6):
        fx = torch.rand(batch, 3, 1, 1, generator=gen) * 12.0
        fy = torch.rand(batch, 3, 1, 1, generator=gen) * 12.0
        ph = torch.rand(batch, 3, 1, 1, generator=gen) * (2.0 * math.pi)
        amp = torch.rand(batch, 3, 1, 1, generator=gen)
        img = img + amp * torch.sin(2.0 * math.pi * (fx * xx + fy * yy) + ph)
    img = img / img.abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-6)
    return img.contiguous()


def make_batch(rows: int, cols: int, num_sources: int, batch: int = 1, seed: int = 0,
               smooth: bool = False, pose_jitter: float = 0.0) -> Dict[str, object]:
    """Return a DataLoader-style batch dict of synthetic frames and cameras.

    ``pose_jitter`` > 0 perturbs the angle and translation of every batch element
    independently (relative magnitude), so that batched kernels are exercised with a
    different homography family per element.
    """
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)

    def frame():
        if smooth:
            return _smooth_image(gen, batch, rows, cols)
        return (torch.rand(batch, 3, rows, cols, generator=gen) * 2.0 - 1.0).contiguous()

    left = frame()
    rights: List[torch.Tensor] = [frame() for _ in range(num_sources)]

    K = torch.eye(4, dtype=torch.float32)
    K[0, 0] = 0.8 * cols
    K[1, 1] = 0.8 * cols
    K[0, 2] = (cols - 1) / 2.0
    K[1, 2] = (rows - 1) / 2.0
    K = K.view(1, 1, 4, 4).repeat(batch, 1, 1, 1).contiguous()

    poses = []
    for i in range(num_sources):
        ang = 0.03 * (i + 1)
        sign = 1.0 if i % 2 == 0 else -1.0
        T = torch.eye(4, dtype=torch.float32)
        c, s = math.cos(ang), math.sin(ang)
        T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c
        T[0, 3], T[1, 3], T[2, 3] = sign * 0.5 * (i + 1), 0.05, 0.02
        Tb = T.view(1, 1, 4, 4).repeat(batch, 1, 1, 1).contiguous()
        if pose_jitter > 0.0:
            for b in range(batch):
                j = (torch.rand(4, generator=gen) * 2.0 - 1.0) * pose_jitter
                a = ang * (1.0 + float(j[0]))
                cb, sb = math.cos(a), math.sin(a)
                Tb[b, 0, 0, 0], Tb[b, 0, 0, 2], Tb[b, 0, 2, 0], Tb[b, 0, 2, 2] = cb, sb, -sb, cb
                Tb[b, 0, 0, 3] *= 1.0 + float(j[1])
                Tb[b, 0, 1, 3] *= 1.0 + float(j[2])
                Tb[b, 0, 2, 3] *= 1.0 + float(j[3])
        poses.append(Tb)

    return {"left_image": left, "right_image": rights, "K": K, "T_right_in_left": poses,
            "left_filename": ["synthetic"] * batch,
            "right_filename": [["synthetic"] * batch for _ in range(num_sources)]}
