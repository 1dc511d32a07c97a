#!/usr/bin/env python3
"""Pin multi_view_stereonet_amd/datasets.py to the reference's readers.

Build-container only.  Writes a miniature GTA-SfM tree and a miniature DeMoN tree in the on-disk formats the
reference reads (jpg + .npy depth + intrinsics.txt / poses.txt / cam.txt), imports the REFERENCE readers and its
evaluation transforms from /root/reference and records every sample they yield.  The fixture holds the dataset files
themselves (bytes) next to the reference's samples, so the test replays the same files through this build's readers.

The reference's dataset modules import torchvision and pyquaternion, which this image lacks.  Stand-ins are installed
for the few entry points the evaluation path touches -- transforms.Compose / Lambda and functional.resize /
to_tensor / normalize, written to torchvision's documented semantics for PIL inputs (resize: PIL bilinear to
(rows, cols); to_tensor: HWC uint8 -> CHW float / 255, float arrays unscaled, 2-D arrays -> 1xHxW; normalize:
(x - mean) / std).  What the fixture pins is therefore the readers' own logic: split-file parsing, the shuffle on read,
calibration look-up by image id, the -0.5 px principal-point fix, pose composition, the DeMoN neighbour windows, the
sample dictionary, and the order in which the transforms touch K -- not torchvision's resampling kernel.

    python tests/golden/make_dataset_golden.py        ->  tests/golden/g10_datasets.npz
"""
import io
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def install_stand_ins():
    import importlib.machinery

    def module(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)     # some importers probe find_spec()
        return m

    tv = module("torchvision")
    tr = module("torchvision.transforms")
    fn = module("torchvision.transforms.functional")

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t_ in self.transforms:
                x = t_(x)
            return x

    class Lambda:
        def __init__(self, lambd):
            self.lambd = lambd

        def __call__(self, x):
            return self.lambd(x)

    def resize(img, size):
        return img.resize((size[1], size[0]), Image.BILINEAR)

    def to_tensor(pic):
        if isinstance(pic, np.ndarray):
            if pic.ndim == 2:
                pic = pic[:, :, None]
            t_ = torch.from_numpy(np.ascontiguousarray(pic.transpose((2, 0, 1))))
            return t_.float().div(255) if t_.dtype == torch.uint8 else t_
        arr = np.array(pic.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(arr.transpose((2, 0, 1)).copy()).float().div(255)

    def normalize(tensor, mean, std):
        m = torch.as_tensor(mean, dtype=tensor.dtype).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=tensor.dtype).view(-1, 1, 1)
        return (tensor - m) / s

    tr.Compose, tr.Lambda, tr.functional = Compose, Lambda, fn
    fn.resize, fn.to_tensor, fn.normalize = resize, to_tensor, normalize
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn})
    pq = module("pyquaternion")
    pq.Quaternion = object
    sys.modules["pyquaternion"] = pq


def write_trees(root):
    """Deterministic miniature datasets; returns {relative path: bytes} of everything written."""
    rng = np.random.default_rng(1234)

    def jpg(path, rows, cols):
        Image.fromarray(rng.integers(0, 255, (rows, cols, 3), dtype=np.uint8), "RGB").save(path, quality=92)

    def pose(i, scale):
        a = 0.05 * i
        T = np.eye(4, dtype=np.float32)
        T[0, 0], T[0, 2], T[2, 0], T[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        T[:3, 3] = [scale * i, 0.02 * i, -0.01 * i]
        return T

    # GTA-SfM: <scene>/<seq>/images/<id>.jpg, depth/<id>.npy, intrinsics.txt, poses.txt (one header line each)
    for scene, seq, ids in (("sceneA", "0000", (0, 1, 2, 5)), ("sceneB", "0003", (10, 11, 12))):
        d = os.path.join(root, "gta", scene, seq)
        os.makedirs(os.path.join(d, "images"))
        os.makedirs(os.path.join(d, "depth"))
        with open(os.path.join(d, "intrinsics.txt"), "w") as fk, open(os.path.join(d, "poses.txt"), "w") as fp:
            fk.write("id fx 0 cx 0 fy cy 0 0 1\n")
            fp.write("id T_cam_in_world (row major)\n")
            for i in ids:
                K = [40.0 + i, 0, 24.0, 0, 41.0 + i, 16.0, 0, 0, 1]
                fk.write(str(i) + " " + " ".join(repr(float(v)) for v in K) + "\n")
                fp.write(str(i) + " " + " ".join(repr(float(v)) for v in pose(i, 0.3).reshape(-1)) + "\n")
                jpg(os.path.join(d, "images", f"{i:04d}.jpg"), 32, 48)
                np.save(os.path.join(d, "depth", f"{i:04d}.npy"), rng.uniform(1.0, 20.0, (32, 48)).astype(np.float32))
    with open(os.path.join(root, "gta_split.txt"), "w") as f:
        f.write("sceneA/0000/images/0001.jpg sceneA/0000/images/0000.jpg sceneA/0000/images/0002.jpg\n"
                "sceneA/0000/images/0002.jpg sceneA/0000/images/0001.jpg sceneA/0000/images/0005.jpg\n"
                "sceneB/0003/images/0011.jpg sceneB/0003/images/0010.jpg sceneB/0003/images/0012.jpg\n"
                "sceneA/0000/images/0005.jpg sceneA/0000/images/0002.jpg sceneA/0000/images/0000.jpg\n")
    # DeMoN: <scene>/NNNNNNN.jpg + .npy, cam.txt (3x3), poses.txt (3x4 world-in-camera per image)
    for scene, n in (("sun3d_a", 5), ("rgbd_b", 3)):
        d = os.path.join(root, "demon", scene)
        os.makedirs(d)
        np.savetxt(os.path.join(d, "cam.txt"), np.array([[50.0, 0, 23.5], [0, 52.0, 15.5], [0, 0, 1]]))
        rows = []
        for i in range(n):
            rows.append(np.linalg.inv(pose(i, 0.2))[:3, :].reshape(-1))
            jpg(os.path.join(d, f"{i:07d}.jpg"), 32, 48)
            np.save(os.path.join(d, f"{i:07d}.npy"), rng.uniform(0.6, 9.0, (32, 48)).astype(np.float64))
        np.savetxt(os.path.join(d, "poses.txt"), np.array(rows))
    with open(os.path.join(root, "demon", "test.txt"), "w") as f:
        f.write("sun3d_a\nrgbd_b\n")
    files = {}
    for base, _, names in os.walk(root):
        for nme in names:
            p = os.path.join(base, nme)
            files[os.path.relpath(p, root)] = open(p, "rb").read()
    return files


def pack_sample(d, key, sample, root):
    d[key + ":left_image"] = sample["left_image"].numpy()
    d[key + ":K"] = sample["K"].numpy().copy()      # (the reference's K tensor aliases the dataset's own array)
    d[key + ":left_depthmap_true"] = sample["left_depthmap_true"].numpy()
    d[key + ":left_filename"] = np.array(os.path.relpath(sample["left_filename"], root))
    rf = sample["right_filename"]
    d[key + ":right_filename"] = np.array([os.path.relpath(r, root) for r in ([rf] if isinstance(rf, str) else rf)])
    d[key + ":num_right"] = np.int64(len(sample["right_image"]))
    for i in range(len(sample["right_image"])):
        d[f"{key}:right_image_{i}"] = sample["right_image"][i].numpy()
        d[f"{key}:T_{i}"] = sample["T_right_in_left"][i].numpy()
        d[f"{key}:right_depthmap_true_{i}"] = sample["right_depthmap_true"][i].numpy()


def main():
    install_stand_ins()
    sys.path.insert(0, "/root/reference")
    # the reference's `datasets` directory has no __init__.py (a namespace package) and an installed package of the
    # same name would win the import: register the directory as the package explicitly
    pkg = types.ModuleType("datasets")
    pkg.__path__ = ["/root/reference/datasets"]
    sys.modules["datasets"] = pkg
    from datasets import multi_view_stereo_dataset as ref_mvsd
    from datasets import gta_sfm_dataset as ref_gta
    from datasets import demon_dataset as ref_demon
    root = tempfile.mkdtemp(prefix="mvsn_ds_")
    try:
        files = write_trees(root)
        d = {"file_names": np.array(sorted(files))}
        for i, name in enumerate(sorted(files)):
            d[f"file_{i}"] = np.frombuffer(files[name], dtype=np.uint8)
        params = {"size": [16, 24]}
        # GTA-SfM: the shuffle on read uses numpy's global generator
        np.random.seed(7)
        data = ref_gta.GTASfMMultiViewStereoDataset(os.path.join(root, "gta"), os.path.join(root, "gta_split.txt"),
                                                    transform=ref_mvsd.get_testing_transforms(params),
                                                    load_groundtruth_depthmaps=True)
        d["gta:len"] = np.int64(len(data))
        for i in range(len(data)):
            pack_sample(d, f"gta:{i}", data[i], root)
        # the same files through a DataLoader-collated batch of the pruned, untransformed-size dataset
        np.random.seed(3)
        pruned = ref_gta.GTASfMMultiViewStereoDataset(os.path.join(root, "gta"), os.path.join(root, "gta_split.txt"),
                                                      num_images=2, transform=ref_mvsd.get_testing_transforms({"size": [32, 48]}),
                                                      load_groundtruth_depthmaps=True)
        d["gta_pruned:len"] = np.int64(len(pruned))
        for i in range(len(pruned)):
            pack_sample(d, f"gta_pruned:{i}", pruned[i], root)
        # DeMoN: python's global random for the shuffle; two neighbour-window sizes.  Read at the frames' own size, as
        # the reference's evaluation does (DeMoN frames are 640x480 = the yaml's size): its ResizeImageStereo scales
        # sample["K"] IN PLACE (datasets/multi_view_stereo_dataset.py:197-199) and DeMoNDataset hands every sample of
        # a scene the same K array (datasets/demon_dataset.py:72-74, :101), so with a real resize K would shrink again
        # on every access -- a latent defect that an identity resize never shows.  Recorded separately below.
        for nr in (1, 2):
            random.seed(11 + nr)
            dm = ref_demon.DeMoNDataset(os.path.join(root, "demon"), "test.txt", num_right_images=nr,
                                        transform=ref_mvsd.get_testing_transforms({"size": [32, 48]}))
            d[f"demon{nr}:len"] = np.int64(len(dm))
            for i in range(len(dm)):
                pack_sample(d, f"demon{nr}:{i}", dm[i], root)
        random.seed(5)
        dm = ref_demon.DeMoNDataset(os.path.join(root, "demon"), "test.txt", num_right_images=1,
                                    transform=ref_mvsd.get_testing_transforms(params))
        pack_sample(d, "demon_resized:0", dm[0], root)            # first access of the scene: K scaled once (correct)
        other = next(i for i in range(1, len(dm)) if os.path.dirname(dm.samples[i]["left_filename"]) ==
                     os.path.dirname(dm.samples[0]["left_filename"]))
        d["demon_resized:other_index"] = np.int64(other)
        d["demon_resized:K_other"] = dm[other]["K"].numpy()       # same scene, next access: scaled twice (the defect)
        np.savez_compressed(os.path.join(HERE, "g10_datasets.npz"), **d)
        print("g10_datasets.npz ok:", len(files), "files,", int(d["gta:len"]), "+", int(d["demon1:len"]), "+",
              int(d["demon2:len"]), "samples")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
