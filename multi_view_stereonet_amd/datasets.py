"""On-disk dataset readers and the evaluation-time transforms.

Same classes, constructor arguments and sample dictionaries as the reference's readers, so configs
2-4 of BASELINE.json can run on the real data if it is ever mounted (it is not here: the tests build
miniature datasets in the same on-disk formats).  torchvision is not available in this image, so the
three transforms the evaluation path uses are written directly on PIL / torch:

* ``read_images``                       <- datasets/multi_view_stereo_dataset.py:16-44
* ``ResizeImageStereo``                 <- :175-208  (bilinear PIL resize; K rows scaled by the size ratio)
* ``to_tensor_stereo`` / ``normalize_stereo`` / ``get_testing_transforms`` <- :46-48, :100-124, :68-98
* ``MultiViewStereoDataset``            <- :227-328
* ``GTASfMMultiViewStereoDataset``      <- datasets/gta_sfm_dataset.py:341-434 (intrinsics.txt / poses.txt,
                                           the -0.5 px principal-point fix :400-411, depth/<id>.npy)
* ``DeMoNDataset``                      <- datasets/demon_dataset.py:18-161 (cam.txt, poses.txt, neighbour choice)

Pinned to the reference's own readers: tests/golden/make_dataset_golden.py runs them (with stand-ins for the
torchvision entry points they call) over miniature trees and records every sample; tests/test_datasets_cpu.py replays
the same files through this module (g10_datasets.npz), next to known-answer tests on synthetic files.
"""
import glob
import os
import random
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.utils.data as tud
from PIL import Image


def read_images(image_file: str):
    """Each line: ``left.jpg right0.jpg ... rightN.jpg`` (paths relative to the data root)."""
    left, right = [], []
    with open(image_file, "r") as f:
        for line in f:
            tok = line.split()
            if tok:
                left.append(tok[0])
                right.append(tok[1:])
    return left, right


# ---- transforms --------------------------------------------------------------------------------
def _image_to_tensor(img) -> torch.Tensor:
    """PIL RGB -> float32 (3,H,W) in [0,1]; 2-D numpy -> (1,H,W) unchanged in value (ToTensor semantics)."""
    if isinstance(img, Image.Image):
        arr = np.asarray(img.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[None]
    return torch.from_numpy(np.ascontiguousarray(arr))


class ResizeImageStereo:
    """Resize the reference and source images; scale K's first two rows by the size ratio.  Ground
    truth is NOT resized (the reference leaves it alone)."""

    def __init__(self, rows: int, cols: int):
        self.rows, self.cols = rows, cols

    def __call__(self, sample: Dict) -> Dict:
        in_cols, in_rows = sample["left_image"].size
        size = (self.cols, self.rows)
        sample["left_image"] = sample["left_image"].resize(size, Image.BILINEAR)
        sample["right_image"] = [im.resize(size, Image.BILINEAR) for im in sample["right_image"]]
        sample["K"] = np.array(sample["K"], dtype=np.float32, copy=True)
        sample["K"][0, :] *= float(self.cols) / in_cols
        sample["K"][1, :] *= float(self.rows) / in_rows
        return sample


def to_tensor_stereo(sample: Dict) -> Dict:
    sample["left_image"] = _image_to_tensor(sample["left_image"])
    sample["right_image"] = [_image_to_tensor(im) for im in sample["right_image"]]
    sample["K"] = _image_to_tensor(np.asarray(sample["K"], dtype=np.float32))                      # (1,4,4)
    sample["T_right_in_left"] = [_image_to_tensor(np.asarray(T, dtype=np.float32)) for T in sample["T_right_in_left"]]
    if "left_depthmap_true" in sample:
        sample["left_depthmap_true"] = _image_to_tensor(np.asarray(sample["left_depthmap_true"], dtype=np.float32))
        sample["right_depthmap_true"] = [_image_to_tensor(np.asarray(d, dtype=np.float32))
                                         for d in sample["right_depthmap_true"]]
    return sample


def normalize_stereo(sample: Dict, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> Dict:
    m = torch.tensor(mean).view(3, 1, 1)
    s = torch.tensor(std).view(3, 1, 1)
    sample["left_image"] = (sample["left_image"] - m) / s
    sample["right_image"] = [(im - m) / s for im in sample["right_image"]]
    return sample


class Compose:
    def __init__(self, steps: Sequence[Callable]):
        self.steps = list(steps)

    def __call__(self, sample):
        for step in self.steps:
            sample = step(sample)
        return sample


def get_testing_transforms(params: Dict) -> Compose:
    """Resize to params["size"] = [rows, cols], to tensors, normalise to [-1, 1]."""
    return Compose([ResizeImageStereo(params["size"][0], params["size"][1]), to_tensor_stereo, normalize_stereo])


# ---- readers -------------------------------------------------------------------------------------
class MultiViewStereoDataset(tud.Dataset):
    """One reference image + N source images per sample, listed in a split file."""

    def __init__(self, data_dir: str, image_file: str, num_images: int = 0, transform: Optional[Callable] = None,
                 load_groundtruth_depthmaps: bool = False, shuffle_on_read: bool = True, seed: Optional[int] = None):
        super().__init__()
        self.data_dir, self.image_file, self.transform = data_dir, image_file, transform
        self.load_groundtruth_depthmaps = load_groundtruth_depthmaps
        self.left_filenames, self.right_filenames = read_images(image_file)
        if shuffle_on_read:   # the reference shuffles on read with the global numpy RNG
            rng = np.random if seed is None else np.random.RandomState(seed)
            perm = rng.permutation(len(self.left_filenames))
            self.left_filenames = [self.left_filenames[i] for i in perm]
            self.right_filenames = [self.right_filenames[i] for i in perm]
        if num_images > 0:
            self.left_filenames = self.left_filenames[:num_images]
            self.right_filenames = self.right_filenames[:num_images]

    def get_calibration(self, idx: int):
        raise NotImplementedError

    def get_groundtruth_depthmap(self, image_filename: str):
        raise NotImplementedError

    def __len__(self):
        return len(self.left_filenames)

    def __getitem__(self, idx):
        idx = int(idx)
        left_filename = os.path.join(self.data_dir, self.left_filenames[idx])
        right_filenames = [os.path.join(self.data_dir, r) for r in self.right_filenames[idx]]
        for f in [left_filename] + right_filenames:
            if not os.path.exists(f):
                raise AssertionError(f"missing image {f}")
        K, T_right_in_left = self.get_calibration(idx)
        # "right_filename" is the LAST source image's path, not the list: the reference builds the dict from its
        # loop variable (datasets/multi_view_stereo_dataset.py:303-311) and its callers hash that one string
        # (multi_view_stereonet_utils.py:302-304)
        sample = {"left_filename": left_filename, "right_filename": right_filenames[-1],
                  "left_image": Image.open(left_filename), "right_image": [Image.open(f) for f in right_filenames],
                  "K": K, "T_right_in_left": T_right_in_left}
        if self.load_groundtruth_depthmaps:
            sample["left_depthmap_true"] = self.get_groundtruth_depthmap(left_filename)
            sample["right_depthmap_true"] = [self.get_groundtruth_depthmap(f) for f in right_filenames]
        return self.transform(sample) if self.transform else sample


def _rows_by_id(path: str, width: int) -> Dict[int, np.ndarray]:
    """``id v0 ... v{width-1}`` rows after one header line."""
    data = np.loadtxt(path, skiprows=1, dtype=np.float32, ndmin=2)
    return {int(r[0]): r[1:1 + width] for r in data}


class GTASfMMultiViewStereoDataset(MultiViewStereoDataset):
    """<scene>/<seq>/images/<id>.jpg, <scene>/<seq>/depth/<id>.npy, intrinsics.txt / poses.txt per sequence."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.left_K, self.left_poses, self.right_poses = [], [], []
        cache: Dict[str, tuple] = {}
        for left_name, right_names in zip(self.left_filenames, self.right_filenames):
            tok = left_name.split(os.path.sep)
            seq = os.path.join(self.data_dir, tok[0], tok[1])
            if seq not in cache:
                cache[seq] = (_rows_by_id(os.path.join(seq, "intrinsics.txt"), 9),
                              _rows_by_id(os.path.join(seq, "poses.txt"), 16))
            Ks, poses = cache[seq]
            image_id = int(os.path.splitext(tok[-1])[0])
            K3 = Ks[image_id].reshape(3, 3).copy()
            # the simulated principal point is cols/2, rows/2; with pixel centres at integers the image
            # centre is (cols-1)/2, (rows-1)/2
            K3[0, 2] -= 0.5
            K3[1, 2] -= 0.5
            self.left_K.append(K3)
            self.left_poses.append(poses[image_id].reshape(4, 4))
            self.right_poses.append([poses[int(os.path.splitext(r.split(os.path.sep)[-1])[0])].reshape(4, 4)
                                     for r in right_names])

    def get_calibration(self, idx: int):
        K = np.eye(4, dtype=np.float32)
        K[:3, :3] = self.left_K[idx]
        inv_left = np.linalg.inv(self.left_poses[idx])
        return K, [np.dot(inv_left, P).astype(np.float32) for P in self.right_poses[idx]]

    def get_groundtruth_depthmap(self, image_filename: str):
        tok = image_filename.split(os.path.sep)
        tok[-2] = "depth"
        tok[-1] = tok[-1].replace("jpg", "npy")
        return np.load(os.path.sep.join(tok))


class DeMoNDataset(tud.Dataset):
    """<scene>/NNNNNNN.jpg + .npy depth, cam.txt (3x3), poses.txt (one 3x4 world-in-camera per image)."""

    def __init__(self, data_dir: str, input_file: str, num_right_images: int = 1, num_left_images: int = 0,
                 transform: Optional[Callable] = None, shuffle_on_read: bool = True, seed: Optional[int] = None):
        self.data_dir, self.input_file, self.transform = data_dir, input_file, transform
        self.num_right_images, self.num_left_images = num_right_images, num_left_images
        with open(os.path.join(data_dir, input_file), "r") as f:
            self.scenes = sorted(os.path.join(data_dir, s.strip()) for s in f if s.strip())
        self.samples = self.generate_samples(num_right_images)
        if shuffle_on_read:
            (random if seed is None else random.Random(seed)).shuffle(self.samples)
        if num_left_images > 0:
            self.samples = self.samples[:num_left_images]
        self.left_filename_to_idx = {s["left_filename"]: i for i, s in enumerate(self.samples)}

    @staticmethod
    def neighbour_indices(left_idx: int, num_images: int, num_right: int) -> List[int]:
        """The window of num_right+1 frames around left_idx (clamped at the sequence ends), minus itself."""
        demi = (num_right + 1) // 2
        if left_idx < demi:
            shifts = list(range(0, num_right + 1))
            shifts.pop(left_idx)
        elif left_idx >= num_images - demi:
            shifts = list(range(num_images - (num_right + 1), num_images))
            shifts.pop(left_idx - num_images)
        else:
            shifts = list(range(left_idx - demi, left_idx + (num_right + 2) // 2))
            shifts.pop(demi)
        return shifts

    def generate_samples(self, num_right: int):
        samples = []
        for scene in self.scenes:
            K = np.eye(4, dtype=np.float32)
            K[:3, :3] = np.genfromtxt(os.path.join(scene, "cam.txt")).astype(np.float32).reshape(3, 3)
            inv_poses = np.genfromtxt(os.path.join(scene, "poses.txt")).astype(np.float32).reshape(-1, 12)
            images = sorted(glob.glob(os.path.join(scene, "*.jpg")))
            if len(images) < num_right + 1:
                continue
            # float32 rows under an integer bottom row promote to float64: the reference composes the relative poses
            # in double and rounds once at the end (datasets/demon_dataset.py:96, :114-116)
            bottom = np.array([[0, 0, 0, 1]])
            world_in = [np.concatenate((p.reshape(3, 4), bottom), 0) for p in inv_poses]
            for li, left_filename in enumerate(images):
                shifts = self.neighbour_indices(li, len(images), num_right)
                assert len(shifts) == num_right
                sample = {"K": K, "left_filename": left_filename,
                          "left_depthmap_true_filename": os.path.splitext(left_filename)[0] + ".npy",
                          "right_filename": [images[r] for r in shifts],
                          "right_depthmap_true_filename": [os.path.splitext(images[r])[0] + ".npy" for r in shifts],
                          "T_right_in_left": [(world_in[li] @ np.linalg.inv(world_in[r])).astype(np.float32)
                                              for r in shifts]}
                samples.append(sample)
        return samples

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        raw = self.samples[int(idx)]
        sample = {"left_filename": raw["left_filename"], "right_filename": raw["right_filename"],
                  "left_image": Image.open(raw["left_filename"]),
                  "right_image": [Image.open(f) for f in raw["right_filename"]],
                  "K": raw["K"], "T_right_in_left": raw["T_right_in_left"],
                  "left_depthmap_true": np.load(raw["left_depthmap_true_filename"]).astype(np.float32),
                  "right_depthmap_true": [np.load(f).astype(np.float32) for f in raw["right_depthmap_true_filename"]]}
        return self.transform(sample) if self.transform else sample
