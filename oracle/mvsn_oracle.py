"""CPU oracle for the MultiViewStereoNet plane-sweep forward.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (torch CPU, fp32) of the algorithm the
reference implements with nn.Modules.  It exists to *check* the HIP path; the product never
imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.

Pinning: the reference ships no tests or golden vectors for this path,
so the oracle is pinned against fixtures GENERATED from the reference itself in the build
container (``tests/golden/make_golden.py`` imports /root/reference and records inputs,
intermediates and outputs).  ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it follows (paths relative to /root/reference):

  feature_network            multi_view_stereonet/multi_view_stereonet.py:78-129, utils/resnet.py:93-109
  idepth_samples             multi_view_stereonet.py:131-165, stereo/image_predictor.py:120-209
  plane_sweep_homographies   multi_view_stereonet.py:167-194, stereo/image_predictor.py:400-461
  homography_warp            multi_view_stereonet.py:205-235, stereo/image_predictor.py:470-523
  feature_refiner            multi_view_stereonet.py:424-440
  incremental_feature_volume multi_view_stereonet.py:247-300
  cost_volume_filter         multi_view_stereonet.py:341-353
  soft_argmin                multi_view_stereonet.py:486-492
  idepth_refiner             multi_view_stereonet.py:468-484
  upsample / upsample_mask   multi_view_stereonet.py:372-396
  forward                    multi_view_stereonet.py:538-695

Weights are passed as a flat ``{name: tensor}`` dict with the reference's state_dict keys.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]
GN_GROUPS = 4
GN_EPS = 1e-5
LRELU = 0.2


# --------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------
def _conv(w: Weights, name: str, x: torch.Tensor, stride: int = 1, dilation: int = 1) -> torch.Tensor:
    weight = w[name + ".weight"]
    bias = w.get(name + ".bias")
    k = weight.shape[-1]
    pad = dilation * (k // 2)
    if weight.dim() == 5:
        return F.conv3d(x, weight, bias, stride=stride, padding=pad, dilation=dilation)
    return F.conv2d(x, weight, bias, stride=stride, padding=pad, dilation=dilation)


def _gn_act(w: Weights, name: str, x: torch.Tensor) -> torch.Tensor:
    """GroupNorm(4 groups of 8 channels, eps 1e-5, biased variance) then LeakyReLU(0.2)."""
    y = F.group_norm(x, GN_GROUPS, w[name + ".weight"], w[name + ".bias"], GN_EPS)
    return F.leaky_relu(y, LRELU)


def _res_block(w: Weights, name: str, x: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """x + LReLU(GN(conv3x3_dilated(x)))  -- one conv, no trailing activation (utils/resnet.py:93-109)."""
    return x + _gn_act(w, name + ".bn1", _conv(w, name + ".conv1", x, dilation=dilation))


# --------------------------------------------------------------------------------------
# a2: feature network
# --------------------------------------------------------------------------------------
def feature_network(w: Weights, prefix: str, image: torch.Tensor) -> List[torch.Tensor]:
    """[image, c0, c1, c2, features]; c* are raw stride-2 5x5 conv outputs (no norm/activation)."""
    pyr = [image]
    x = image
    for i in range(3):
        x = _conv(w, f"{prefix}.conv{i}", x, stride=2)
        pyr.append(x)
    x = _conv(w, f"{prefix}.conv3", x, stride=2)
    for i in range(6):
        x = _res_block(w, f"{prefix}.res{i}", x)
    pyr.append(_conv(w, f"{prefix}.conv_final", x))
    return pyr


# --------------------------------------------------------------------------------------
# a3: idepth samples
# --------------------------------------------------------------------------------------
def _pixel_grid(rows: int, cols: int, device) -> torch.Tensor:
    """(3, rows*cols) homogeneous pixel centres, x fastest."""
    ys, xs = torch.meshgrid(torch.arange(rows, device=device, dtype=torch.float32),
                            torch.arange(cols, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(rows * cols, device=device)], 0)


def disparity_to_idepth(K: torch.Tensor, T_right_in_left: torch.Tensor, disparity: float,
                        rows: int, cols: int) -> torch.Tensor:
    """Inverse depth that moves each pixel ``disparity`` px along its epipolar line. (B, rows*cols)."""
    grid = _pixel_grid(rows, cols, K.device)[None]                       # (1,3,P)
    Kinv = torch.linalg.inv(K)
    T_lr = torch.linalg.inv(T_right_in_left)
    M = K[:, :3, :3] @ (T_lr[:, :3, :3] @ Kinv[:, :3, :3])              # K R K^-1
    Kt = (K @ T_lr)[:, :3, 3]                                            # (B,3)

    inf = M @ grid                                                       # (B,3,P)
    inf_xy = inf[:, :2] / inf[:, 2:3]
    far = M @ (grid * 1e2) + Kt[:, :, None]
    far_xy = far[:, :2] / far[:, 2:3]

    diff = far_xy - inf_xy
    norm = diff.pow(2).sum(1).sqrt()                                     # (B,P)
    line = diff / (norm[:, None] + 1e-6)
    degenerate = norm < 1e-6

    wz = M[:, 2, 0, None] * grid[:, 0] + M[:, 2, 1, None] * grid[:, 1] + M[:, 2, 2, None]
    A0 = Kt[:, 0, None] - Kt[:, 2, None] * (inf_xy[:, 0] + disparity * line[:, 0])
    A1 = Kt[:, 1, None] - Kt[:, 2, None] * (inf_xy[:, 1] + disparity * line[:, 1])
    b0 = wz * disparity * line[:, 0]
    b1 = wz * disparity * line[:, 1]
    idepth = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1)
    return (~degenerate).float() * idepth


def idepth_samples(T_right_in_left: torch.Tensor, K: torch.Tensor, rows: int, cols: int,
                   num: int) -> torch.Tensor:
    """(B, num) linear samples from 0 (plane at infinity) to the per-view maximum idepth."""
    m = disparity_to_idepth(K, T_right_in_left, float(num - 1), rows, cols)
    m = (m > 0).float() * m
    top = m.sum(1) / (m > 0).sum(1)
    top = torch.where(top > 2.0, torch.full_like(top, 2.0), top)
    tz = T_right_in_left[:, 2, 3]
    behind = (1.0 / top) < tz
    top = torch.where(behind, 1.0 / tz, top)
    delta = top / (num - 1)
    steps = torch.arange(0.0, num, device=K.device)
    return steps[None, :] * delta[:, None] + 0.0


# --------------------------------------------------------------------------------------
# a4: homographies
# --------------------------------------------------------------------------------------
def plane_sweep_homographies(T_right_in_left: torch.Tensor, K: torch.Tensor,
                             idepths: torch.Tensor) -> torch.Tensor:
    """H[b,d] = K (R + t*idepth[b,d] e3^T) K^-1, (R,t) = inverse(T_right_in_left). (B,n,3,3)."""
    T_lr = torch.linalg.inv(T_right_in_left)
    K3 = K[:, :3, :3]
    K3inv = torch.linalg.inv(K3)
    R = T_lr[:, :3, :3]
    t = T_lr[:, :3, 3]
    B, n = idepths.shape
    core = R[:, None].repeat(1, n, 1, 1)
    core[:, :, :, 2] = core[:, :, :, 2] + t[:, None, :] * idepths[:, :, None]
    return K3[:, None] @ (core @ K3inv[:, None])


# --------------------------------------------------------------------------------------
# a5: homography warp (explicit bilinear gather; same arithmetic order as grid_sample)
# --------------------------------------------------------------------------------------
def warp_coordinates(H: torch.Tensor, rows: int, cols: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """For H (N,3,3): source-pixel coords (ix, iy) as grid_sample un-normalises them, and the
    out-of-image predicate evaluated on the normalised coordinates exactly as
    stereo/image_predictor.py:506-516 does: n = ((p+0.5)*2)/size - 1, mask = |nx|>1 or |ny|>1."""
    grid = _pixel_grid(rows, cols, H.device)[None]
    u = H @ grid                                                         # (N,3,P)
    px = u[:, 0] / u[:, 2]
    py = u[:, 1] / u[:, 2]
    nx = ((px + 0.5) * 2.0) / cols - 1.0
    ny = ((py + 0.5) * 2.0) / rows - 1.0
    mask = (nx.abs() > 1.0) | (ny.abs() > 1.0)
    ix = ((nx + 1.0) * cols - 1.0) / 2.0
    iy = ((ny + 1.0) * rows - 1.0) / 2.0
    return ix, iy, mask


def homography_warp(image: torch.Tensor, H: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """image (B,C,h,w), H (B,n,3,3) -> volume (B,C,n,h,w) with out-of-image voxels zeroed,
    mask (B,n,h,w) bool (True = outside).  Bilinear, clamp-to-edge taps (padding_mode=border,
    align_corners=False)."""
    B, C, rows, cols = image.shape
    n = H.shape[1]
    ix, iy, mask = warp_coordinates(H.reshape(B * n, 3, 3), rows, cols)  # (B*n,P)
    ix = ix.clamp(0.0, cols - 1.0)
    iy = iy.clamp(0.0, rows - 1.0)
    x0 = ix.floor()
    y0 = iy.floor()
    fx = ix - x0
    fy = iy - y0
    x0i = x0.long()
    y0i = y0.long()
    x1i = (x0i + 1).clamp(max=cols - 1)      # weight is exactly 0 whenever the clamp acts
    y1i = (y0i + 1).clamp(max=rows - 1)
    flat = image.reshape(B, 1, C, rows * cols).expand(B, n, C, rows * cols).reshape(B * n, C, rows * cols)

    def tap(yi, xi):
        idx = (yi * cols + xi)[:, None, :].expand(-1, C, -1)
        return torch.gather(flat, 2, idx)

    w00 = ((1.0 - fx) * (1.0 - fy))[:, None]
    w01 = (fx * (1.0 - fy))[:, None]
    w10 = ((1.0 - fx) * fy)[:, None]
    w11 = (fx * fy)[:, None]
    out = tap(y0i, x0i) * w00 + tap(y0i, x1i) * w01 + tap(y1i, x0i) * w10 + tap(y1i, x1i) * w11
    out = out * (~mask).float()[:, None]
    vol = out.reshape(B, n, C, rows, cols).permute(0, 2, 1, 3, 4).contiguous()
    return vol, mask.reshape(B, n, rows, cols)


# --------------------------------------------------------------------------------------
# a7 / a6: feature refiner and the incremental chain
# --------------------------------------------------------------------------------------
def feature_refiner(w: Weights, prefix: str, image: torch.Tensor, feats: torch.Tensor) -> torch.Tensor:
    x = torch.cat([image, feats], 1)
    x = _gn_act(w, prefix + ".bn0", _conv(w, prefix + ".conv0", x))
    x = _res_block(w, prefix + ".res0", x)
    return feats + _conv(w, prefix + ".conv_final", x)


def inv3x3(M: torch.Tensor) -> torch.Tensor:
    """torch.inverse of a plane's homographies as the reference calls it (multi_view_stereonet.py:281): its H family is
    a permuted (D,B,3,3) tensor (:192), so the slice `H[:, d-1]` is CONTIGUOUS for every batch size and ATen's inverse
    takes its transposed-LU shortcut; this oracle's (B,D,3,3) slice is strided for B > 1 and would take ATen's other
    route, which rounds differently (4e-5 of a depth map on noise frames) -- hence the copy."""
    return torch.linalg.inv(M.contiguous())


def incremental_feature_volume(w: Weights, T: torch.Tensor, K_pyr: List[torch.Tensor],
                               right_pyr: List[torch.Tensor], idepths: torch.Tensor,
                               cap: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Viewpoint-compensated source features for every plane: (B,32,D,h4,w4), mask (B,D,h4,w4)."""
    D = idepths.shape[1]
    H0 = plane_sweep_homographies(T, K_pyr[0], idepths[:, :1])
    warped0, _ = homography_warp(right_pyr[0], H0)
    f = feature_network(w, "right_feature_extractor.feature_extractor", warped0[:, :, 0])[-1]

    H = plane_sweep_homographies(T, K_pyr[-1], idepths)
    image_vol, mask_vol = homography_warp(right_pyr[-1], H)
    if cap is not None:
        cap["H_lvl0_plane0"] = H0
        cap["H"] = H
        cap["image_volume"] = image_vol
        cap["plane0_features"] = f
    planes = [f]
    for d in range(1, D):
        H_inc = inv3x3(H[:, d - 1]) @ H[:, d].contiguous()
        moved, _ = homography_warp(planes[-1], H_inc[:, None])
        planes.append(feature_refiner(w, "right_feature_extractor.refiner", image_vol[:, :, d], moved[:, :, 0]))
    vol = torch.stack(planes, 2)
    vol = vol * (~mask_vol).float()[:, None]
    return vol, mask_vol


# --------------------------------------------------------------------------------------
# a9 / a10 / a11 / a12
# --------------------------------------------------------------------------------------
def cost_volume_filter(w: Weights, prefix: str, vol: torch.Tensor) -> torch.Tensor:
    x = vol
    for i in range(4):
        x = _gn_act(w, f"{prefix}.bn{i}", _conv(w, f"{prefix}.conv{i}", x))
    return _conv(w, f"{prefix}.conv4", x)[:, 0]


def soft_argmin(cost: torch.Tensor, idepths: torch.Tensor) -> torch.Tensor:
    """sum_d softmax(-cost)_d * idepth_d  -> (B,1,h,w)."""
    p = torch.softmax(-cost, dim=1)
    return (p * idepths[:, :, None, None]).sum(1, keepdim=True)


DILATIONS = (1, 2, 4, 8, 1, 1)


def idepth_refiner(w: Weights, prefix: str, guide: torch.Tensor, idepth: torch.Tensor) -> torch.Tensor:
    x = torch.cat([guide, idepth], 1)
    x = _gn_act(w, prefix + ".bn0", _conv(w, prefix + ".conv0", x))
    for i, dil in enumerate(DILATIONS):
        x = _res_block(w, f"{prefix}.res{i}", x, dilation=dil)
    return F.relu(idepth + _conv(w, prefix + ".conv_final", x))


def upsample(x: torch.Tensor, size) -> torch.Tensor:
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


def upsample_mask(mask: torch.Tensor, size) -> torch.Tensor:
    return upsample(mask.float(), size) > 0.5


# --------------------------------------------------------------------------------------
# a1: the forward
# --------------------------------------------------------------------------------------
def forward(w: Weights, left_image_pyr: List[torch.Tensor], K_pyr: List[torch.Tensor],
            T_right_in_lefts: List[torch.Tensor], right_image_pyrs: List[List[torch.Tensor]],
            num_idepth_samples: int, do_cost_volume_filter: bool = True,
            do_refiners: Optional[List[bool]] = None, capture: Optional[dict] = None):
    """Same contract as MultiViewStereoNet.forward; optionally records intermediates."""
    do_refiners = [True] * 5 if do_refiners is None else list(do_refiners)
    assert len(K_pyr) == 5 and len(left_image_pyr) == 5
    S = len(T_right_in_lefts)
    D = num_idepth_samples
    left_feats = feature_network(w, "left_feature_extractor", left_image_pyr[0])
    FL = left_feats[-1]
    B, _, h4, w4 = FL.shape
    img_rows4, img_cols4 = left_image_pyr[-1].shape[-2:]

    def refine(level: int, prior: torch.Tensor) -> torch.Tensor:
        if not do_refiners[level]:
            return prior
        fx = K_pyr[level][:, 0, 0].view(-1, 1, 1, 1)
        guide = left_image_pyr[level] if level == 0 else torch.cat([left_image_pyr[level], left_feats[level]], 1)
        return idepth_refiner(w, f"refiner{level}", guide, prior * fx) / fx

    raw_sum = torch.zeros(B, 1, h4, w4)
    ref_sum = torch.zeros(B, 1, h4, w4)
    mask_sum = torch.zeros(B, D, h4, w4)
    for s in range(S):
        T = T_right_in_lefts[s].clone()
        baseline = T[:, :3, 3].pow(2).sum(1).sqrt()
        T[:, :3, 3] = T[:, :3, 3] / baseline[:, None]
        samples = idepth_samples(T, K_pyr[-1], img_rows4, img_cols4, D)
        cap_s = {} if capture is not None else None
        FR, mask = incremental_feature_volume(w, T, K_pyr, right_image_pyrs[s], samples, cap_s)
        cost = (~mask).float()[:, None] * (FL[:, :, None] - FR).abs()
        if do_cost_volume_filter:
            filtered = cost_volume_filter(w, "volume_filter4", cost)
        else:
            filtered = cost.pow(2).sum(1).sqrt()
        raw = soft_argmin(filtered, samples)
        # Reference quirk (multi_view_stereonet.py:613,618-619): with refiner 4 disabled the
        # refined map aliases the raw map and is divided by the baseline twice.
        scale = baseline.view(-1, 1, 1, 1)
        if do_refiners[4]:
            refined = refine(4, raw) / scale
            raw = raw / scale
        else:
            raw = raw / scale / scale
            refined = raw
        raw_sum = raw_sum + raw
        ref_sum = ref_sum + refined
        mask_sum = mask_sum + mask.float()
        if capture is not None:
            cap_s.update(idepth_samples=samples, feature_volume=FR, mask_volume=mask, cost_volume=cost,
                         filtered_cost=filtered)
            capture.setdefault("sources", []).append(cap_s)

    idepth = [None] * 5
    prior = [None] * 5
    masks = [None] * 5
    prior[4] = raw_sum / S
    idepth[4] = ref_sum / S
    masks[4] = (mask_sum / S) > 0.5
    for lvl in (3, 2, 1, 0):
        size = left_image_pyr[lvl].shape[-2:]
        prior[lvl] = upsample(idepth[lvl + 1], size)
        masks[lvl] = upsample_mask(masks[lvl + 1], size)
        idepth[lvl] = refine(lvl, prior[lvl])
    if capture is not None:
        capture["left_features"] = left_feats
    return {"left_idepthmap_pyr": idepth, "left_idepthmap_raw_pyr": prior, "left_idepthmap_mask_pyr": masks}


# ---------------------------------------------------------------------------------------------------
# Two-view consistency ops: multi_view_stereonet/losses.py:42-160 over
# stereo/image_predictor.py:36-118 (DepthmapToPointCloud, PointCloudToPixel) and :525-576 (IDepthmapProjector)
# ---------------------------------------------------------------------------------------------------
def idepthmap_projector(K: torch.Tensor, T_right_in_left: torch.Tensor, left_idepthmap: torch.Tensor):
    """image_predictor.py:538-576: every left pixel (x, y, idepth) -> its normalised pixel coordinate in the right
    image (B,rows,cols,2), its idepth in the right frame (B,1,rows,cols) and the out-of-image mask (B,1,rows,cols)."""
    B, _, rows, cols = left_idepthmap.shape
    Kinv = torch.inverse(K)
    T_left_in_right = torch.inverse(T_right_in_left)
    depth = 1.0 / (left_idepthmap + 1e-6)                                       # :557
    ys, xs = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
    pix = torch.stack([xs.reshape(-1).float(), ys.reshape(-1).float(), torch.ones(rows * cols)], 0)
    pts = depth.reshape(B, 1, -1) * torch.matmul(Kinv[:, :3, :3], pix.unsqueeze(0).expand(B, -1, -1))   # :69-70
    pts = torch.cat([pts, torch.ones(B, 1, rows * cols)], 1)
    right_pts = torch.matmul(T_left_in_right[:, :3, :], pts)                    # :563
    right_idepths = (1.0 / (right_pts[:, 2, :] + 1e-6)).view(left_idepthmap.shape)
    cam = torch.matmul(torch.matmul(K, T_left_in_right)[:, :3, :], pts)         # :104-105
    uv = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + 1e-7)
    uv = uv.view(B, 2, rows, cols).permute(0, 2, 3, 1).clone()
    uv += 0.5
    uv *= 2.0
    uv[..., 0] /= cols
    uv[..., 1] /= rows
    uv -= 1.0
    mask = ((uv[..., 0].abs() > 1.0) | (uv[..., 1].abs() > 1.0)).unsqueeze(1)
    return uv, right_idepths, mask


def _sample(image: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    return F.grid_sample(image, uv, mode="bilinear", padding_mode="border", align_corners=False)


def get_occlusion_mask(K, T_right_in_left, left_idepthmap, right_idepthmap) -> torch.Tensor:
    """losses.py:42-82: 1 where a left pixel is occluded in the right view (reprojected idepth farther than the right
    map's by more than the image's mean absolute difference) or leaves the right image."""
    B = left_idepthmap.shape[0]
    uv, id_prime, invalid = idepthmap_projector(K, T_right_in_left, left_idepthmap)
    id_diff = _sample(right_idepthmap, uv) - id_prime
    thr = id_diff.view(B, -1).abs().mean(dim=1).view(B, 1, 1, 1)
    return (id_diff > thr) | invalid


def left_right_consistency_loss(T_right_in_left, T_left_in_right, K_pyr, left_idepthmap_pyr, left_occlusion_mask_pyr,
                                right_idepthmap_pyr, right_occlusion_mask_pyr) -> torch.Tensor:
    """losses.py:112-160: per level, L1 between the reprojected idepths of one view and the other view's map sampled
    at the reprojected pixels, over pixels unoccluded in both; both directions, summed over levels."""
    loss = torch.zeros(())
    for lvl in range(len(left_idepthmap_pyr)):
        if left_idepthmap_pyr[lvl] is None:
            continue
        for T, a, a_occ, b, b_occ in ((T_right_in_left, left_idepthmap_pyr[lvl], left_occlusion_mask_pyr[lvl],
                                       right_idepthmap_pyr[lvl], right_occlusion_mask_pyr[lvl]),
                                      (T_left_in_right, right_idepthmap_pyr[lvl], right_occlusion_mask_pyr[lvl],
                                       left_idepthmap_pyr[lvl], left_occlusion_mask_pyr[lvl])):
            uv, projected, _ = idepthmap_projector(K_pyr[lvl], T, a)
            keep = ~a_occ & ~(_sample(b_occ.float(), uv) > 0)
            loss = loss + F.l1_loss(projected[keep], _sample(b, uv)[keep])
    return loss
