"""Data layer feeding the samplers and the evaluation scripts (reference data.py:11-371).

Host-side numpy, like the reference: voxel grid file -> min-max normalised occupancy -> (optionally) point
cloud of voxel coordinates -> centroid-centred, unit-radius cloud -> resampled to `num_points`.  The class and
method names, constructor arguments and defaults follow the reference so its entry scripts read the same.

Differences, both deliberate:
  * no Lightning: `PointCloudDataModule` / `PointCloudDataDirectoryModule` are plain objects with the same
    `setup()` / `train_dataloader()` / `val_dataloader()` surface;
  * file format: the reference stores each sample as a deepdish HDF5 file `*.dd` read with
    `dd.io.load(path)['data']` (data.py:176).  deepdish/h5py are not part of this image, so besides `.dd`
    (read through h5py when it is importable) the dataset accepts `.npz` (key `data`) and `.npy` files with
    the same naming scheme; `convert_dd_to_npz` rewrites a directory once on a machine that has h5py.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, TensorDataset

_EXTS = (".dd", ".npz", ".npy")

# ShapeNet synset id -> category name (reference data.py:82-138); the id is the 5th '_' field of a file name
SHAPENET_ID_TO_CATEGORY = {
    "02691156": "airplane", "02747177": "ashcan", "02773838": "bag", "02801938": "basket", "02808440": "bathtub",
    "02818832": "bed", "02828884": "bench", "02843684": "birdhouse", "02871439": "bookshelf", "02876657": "bottle",
    "02880940": "bowl", "02924116": "bus", "02933112": "cabinet", "02942699": "camera", "02946921": "can",
    "02954340": "cap", "02958343": "car", "02992529": "cellular_telephone", "03001627": "chair", "03046257": "clock",
    "03085013": "computer_keyboard", "03207941": "dishwasher", "03211117": "display", "03261776": "earphone",
    "03325088": "faucet", "03337140": "file", "03467517": "guitar", "03513137": "helmet", "03593526": "jar",
    "03624134": "knife", "03636649": "lamp", "03642806": "laptop", "03691459": "loudspeaker", "03710193": "mailbox",
    "03759954": "microphone", "03761084": "microwave", "03790512": "motorcycle", "03797390": "mug", "03928116": "piano",
    "03938244": "pillow", "03948459": "pistol", "03991062": "pot", "04004475": "printer", "04074963": "remote_control",
    "04090263": "rifle", "04099429": "rocket", "04225987": "skateboard", "04256520": "sofa", "04330267": "stove",
    "04379243": "table", "04401088": "telephone", "04460130": "tower", "04468005": "train", "04530566": "vessel",
    "04554684": "washer",
}


def load_sample_file(path: str) -> np.ndarray:
    """The `data` array of one sample file (reference: `dd.io.load(file_path)['data']`, data.py:176,192)."""
    if path.endswith(".npz"):
        with np.load(path) as f:
            return f["data"]
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".dd"):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError(f"{path}: reading deepdish .dd files needs h5py, which is not installed here; convert "
                               "the directory once with shapegen_amd.data.convert_dd_to_npz on a machine that has it") from e
        with h5py.File(path, "r") as f:
            return np.asarray(f["data"])
    raise ValueError(f"unsupported sample file {path}")


def convert_dd_to_npz(src_dir: str, dst_dir: str) -> int:
    """Rewrite every `*.dd` under src_dir as `<same name>.npz` (key `data`) in dst_dir; returns the count."""
    os.makedirs(dst_dir, exist_ok=True)
    n = 0
    for f in sorted(os.listdir(src_dir)):
        if f.endswith(".dd"):
            np.savez_compressed(os.path.join(dst_dir, f[:-3] + ".npz"), data=load_sample_file(os.path.join(src_dir, f)))
            n += 1
    return n


class PointCloudDataModule:
    """reference data.py:11-46: in-memory clouds -> random 80/20 split -> loaders."""

    def __init__(self, point_clouds, batch_size=32, train_val_split=0.8):
        self.point_clouds, self.batch_size, self.train_val_split = point_clouds, batch_size, train_val_split

    def setup(self, stage=None):
        dataset = TensorDataset(torch.FloatTensor(np.asarray(self.point_clouds)))
        train_size = int(self.train_val_split * len(dataset))
        self.train_dataset, self.val_dataset = torch.utils.data.random_split(dataset, [train_size, len(dataset) - train_size])

    def train_dataloader(self):
        return DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=True)

    def val_dataloader(self):
        return DataLoader(self.val_dataset, batch_size=self.batch_size)


class PointCloudDataset(Dataset):
    """reference data.py:48-311.  Returns a float32 tensor: (1,R,R,R) in 'voxels' output mode, (num_points,3)
    in 'point_clouds' output mode."""

    def __init__(self, data_dir, num_points=2048, transform=None, input_mode="voxels", output_mode="voxels",
                 normalize=True, jitter=True, rotate=False, resolution=32, relevant_object_categories=None):
        self.data_dir = data_dir
        self.transform = transform
        self.num_points = num_points
        self.file_list: List[str] = [f for f in os.listdir(data_dir) if f.endswith(_EXTS)]
        self.input_mode = input_mode
        self.output_mode = output_mode
        self.normalize = normalize
        self.jitter = jitter
        self.rotate = rotate
        self.resolution = 32          # the reference ignores its `resolution` argument too (data.py:75)
        self.relevant_object_categories = ["all"] if relevant_object_categories is None else relevant_object_categories
        self.shapenet_id_to_category = SHAPENET_ID_TO_CATEGORY
        self.filter_file_list()

    def filter_file_list(self):
        """data.py:140-152: keep files whose synset id (5th '_' field) maps to a requested category."""
        if self.input_mode != "voxels" or self.relevant_object_categories == ["all"]:
            return
        self.file_list = [f for f in self.file_list
                          if self.shapenet_id_to_category[f.split("_")[4]] in self.relevant_object_categories]

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        file_path = os.path.join(self.data_dir, self.file_list[idx])
        if self.input_mode == "voxels":
            voxels = load_sample_file(file_path)
            self.resolution = voxels.shape[0]
            lo, hi = np.min(voxels), np.max(voxels)
            voxels = np.full_like(voxels, lo) if lo == hi else (voxels - lo) / (hi - lo)
            if self.output_mode == "voxels" and self.transform is None and not any([self.jitter, self.rotate]):
                return torch.FloatTensor(np.expand_dims(voxels, axis=0))
            point_cloud = self.voxel_to_point_cloud(voxels)
        elif self.input_mode == "point_clouds":
            point_cloud = load_sample_file(file_path)
        else:
            raise ValueError("Invalid input_mode for PointCloudDataset")

        if self.transform:
            point_cloud = self.transform(point_cloud)
        if self.rotate:
            point_cloud = self.rotate_around_vertical_axis(self.normalize_point_cloud(point_cloud))
        if self.jitter:
            point_cloud = self.jitter_points(point_cloud)

        if self.output_mode == "voxels":
            output = np.expand_dims(self.point_cloud_to_voxel(point_cloud, self.resolution), axis=0)
        elif self.output_mode == "point_clouds":
            if self.normalize:
                point_cloud = self.normalize_point_cloud(point_cloud)
            output = self.sample_point_cloud(point_cloud, self.num_points)
        else:
            raise ValueError("Invalid output_mode for PointCloudDataset")
        return torch.FloatTensor(output)

    # ------------------------------------------------------------------ deterministic pieces
    @staticmethod
    def voxel_to_point_cloud(voxels, threshold=0.5):
        """data.py:213-218: integer (z,y,x) indices of the occupied voxels, row-major scan order."""
        return np.array(np.where(voxels > threshold)).T

    @staticmethod
    def point_cloud_to_voxel(point_cloud, resolution):
        """data.py:220-228: [-1,1] coordinates -> occupancy, written as grid[z,y,x] from columns (x,y,z)."""
        top = resolution - 1
        cell = np.clip((point_cloud + 1) * top / 2, 0, top).astype(int)          # truncation, after the clip, as the reference does
        x, y, z = cell[:, 0], cell[:, 1], cell[:, 2]
        grid = np.zeros((resolution,) * 3, dtype=np.float32)
        grid[z, y, x] = 1
        return grid

    @staticmethod
    def normalize_point_cloud(point_cloud):
        """data.py:230-238: subtract the centroid, divide by the largest distance from it."""
        point_cloud = point_cloud - np.mean(point_cloud, axis=0)
        return point_cloud / np.max(np.sqrt(np.sum(point_cloud ** 2, axis=1)))

    # ------------------------------------------------------------------ random pieces (python / numpy global RNGs)
    @staticmethod
    def sample_point_cloud(point_cloud, num_points):
        """data.py:240-254: exact size -> unchanged; more -> without replacement (`random.sample`); fewer -> every
        point once, then `np.random.choice` with replacement for the rest."""
        if len(point_cloud) == num_points:
            return point_cloud
        if len(point_cloud) > num_points:
            return point_cloud[random.sample(range(len(point_cloud)), num_points)]
        extra = np.random.choice(len(point_cloud), num_points - len(point_cloud), replace=True)
        return point_cloud[list(range(len(point_cloud))) + extra.tolist()]

    @staticmethod
    def farthest_point_sample(point_cloud, num_points):
        """data.py:256-287 (the reference's own pipeline does not call it: "makes dataloading very slow"): greedy farthest-point subset, first
        point from `np.random.randint`.  `nearest` = squared distance of every point to the subset so far (starts at 1e10)."""
        n = len(point_cloud)
        if n == num_points:
            return point_cloud
        xyz = point_cloud[:, :3]
        nearest = np.full((n,), 1e10)
        chosen = np.empty((num_points,), np.int32)
        nxt = np.random.randint(0, n)
        for k in range(num_points):
            chosen[k] = nxt
            np.minimum(nearest, np.sum((xyz - xyz[nxt, :]) ** 2, axis=-1), out=nearest)
            nxt = np.argmax(nearest, axis=-1)
        return point_cloud[chosen]

    @staticmethod
    def jitter_points(points, sigma=0.01, clip=0.05):
        """data.py:289-295."""
        return np.clip(sigma * np.random.randn(*points.shape), -clip, clip) + points

    @staticmethod
    def rotate_around_vertical_axis(point_cloud):
        """data.py:297-309: right-multiply by a rotation about the y axis by a uniform angle."""
        a = np.random.uniform() * 2 * np.pi
        c, s = np.cos(a), np.sin(a)
        return np.dot(point_cloud, np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]))


class PointCloudDataDirectoryModule:
    """reference data.py:311-371 (`augmentations=False` switches jitter and rotation off)."""

    def __init__(self, data_dir, num_points=2048, batch_size=32, num_workers=4, train_val_split=0.8,
                 file_mode="voxels", output_mode="point_clouds", augmentations=True, normalization=True,
                 relevant_object_categories: Optional[Sequence[str]] = None):
        self.data_dir, self.num_points, self.batch_size, self.num_workers = data_dir, num_points, batch_size, num_workers
        self.train_val_split, self.file_mode, self.output_mode = train_val_split, file_mode, output_mode
        self.augmentations, self.normalization = augmentations, normalization
        self.relevant_object_categories = relevant_object_categories

    def setup(self, stage=None):
        kw = dict(num_points=self.num_points, input_mode=self.file_mode, output_mode=self.output_mode,
                  normalize=self.normalization, relevant_object_categories=self.relevant_object_categories)
        if not self.augmentations:
            kw.update(rotate=False, jitter=False)
        full = PointCloudDataset(self.data_dir, **kw)
        train_size = int(self.train_val_split * len(full))
        self.train_dataset, self.val_dataset = torch.utils.data.random_split(full, [train_size, len(full) - train_size])

    def train_dataloader(self):
        return DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=True, num_workers=self.num_workers)

    def val_dataloader(self):
        return DataLoader(self.val_dataset, batch_size=self.batch_size, num_workers=self.num_workers)
