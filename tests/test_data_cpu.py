"""Data layer: shapegen_amd.data against vectors captured from the reference's data.py
(tests/golden/data.npz, written by `python oracle/make_golden.py data`).  The random draws (python `random`,
numpy global RNG) are replayed from the seeds the capture used, so resampled clouds compare bit-exactly."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import make_golden
from shapegen_amd import data as D


@pytest.fixture()
def sample_dir(tmp_path):
    names, vox = make_golden.write_data_dir(str(tmp_path / "dd"))
    root = tmp_path / "npz"
    os.makedirs(root)
    for n, v in zip(names, vox):
        np.savez(os.path.join(root, n[:-3] + ".npz"), data=v)
    return str(root), [n[:-3] + ".npz" for n in names], vox


def test_voxel_mode_and_category_filter(sample_dir, golden):
    root, names, vox = sample_dir
    g = golden("data.npz")
    ds = D.PointCloudDataset(root, input_mode="voxels", output_mode="voxels", jitter=False, rotate=False)
    order = sorted(range(len(ds)), key=lambda i: ds.file_list[i])
    assert [ds.file_list[i][:-4] for i in order] == [str(f)[:-3] for f in g["vv_files"]]
    out = np.stack([ds[i].numpy() for i in order])
    assert out.dtype == np.float32 and np.array_equal(out, g["vv_out"])
    assert out[1].max() == 1.0 and np.all(out[4] == 0.25)          # min-max normalised; constant grid kept
    tab = D.PointCloudDataset(root, input_mode="voxels", output_mode="voxels", jitter=False, rotate=False,
                              relevant_object_categories=["table"])
    assert sorted(f[:-4] for f in tab.file_list) == [str(f)[:-3] for f in g["table_files"]]


def test_point_cloud_mode_exact_and_resampled(sample_dir, golden):
    root, names, vox = sample_dir
    g = golden("data.npz")
    n0 = int((vox[0] > 0.5).sum())
    dp = D.PointCloudDataset(root, num_points=n0, input_mode="voxels", output_mode="point_clouds", jitter=False, rotate=False)
    pc = dp[dp.file_list.index(names[0])].numpy()
    assert np.array_equal(pc, g["pc_exact"])
    assert abs(np.linalg.norm(pc, axis=1).max() - 1.0) < 1e-6 and np.abs(pc.mean(0)).max() < 1e-6
    for tag, npts in (("more", n0 // 3), ("fewer", n0 + 257)):
        dq = D.PointCloudDataset(root, num_points=npts, input_mode="voxels", output_mode="point_clouds", jitter=False, rotate=False)
        random.seed(11); np.random.seed(11)
        got = dq[dq.file_list.index(names[0])].numpy()
        assert got.shape == (npts, 3) and np.array_equal(got, g[f"pc_{tag}"])


def test_augmented_voxel_output_and_helpers(sample_dir, golden):
    root, names, vox = sample_dir
    g = golden("data.npz")
    da = D.PointCloudDataset(root, input_mode="voxels", output_mode="voxels", jitter=True, rotate=True)
    random.seed(12); np.random.seed(12)
    assert np.array_equal(da[da.file_list.index(names[0])].numpy(), g["aug_voxels"])
    pts = g["helper_pts"]
    assert np.array_equal(D.PointCloudDataset.normalize_point_cloud(pts), g["helper_norm"])
    assert np.array_equal(D.PointCloudDataset.point_cloud_to_voxel(g["helper_norm"], 32), g["helper_vox"])
    np.random.seed(13)
    assert np.array_equal(D.PointCloudDataset.farthest_point_sample(pts, 64), g["helper_fps"])       # golden from the reference's data.py:256-287
    # [z,y,x] scan order of voxel_to_point_cloud == the order utils.voxel_tensor_to_point_clouds reads grids in
    v = np.zeros((4, 4, 4)); v[1, 2, 3] = 1; v[0, 3, 1] = 1
    assert D.PointCloudDataset.voxel_to_point_cloud(v).tolist() == [[0, 3, 1], [1, 2, 3]]


def test_directory_module_split_and_loaders(sample_dir):
    root, names, vox = sample_dir
    torch.manual_seed(24)
    dm = D.PointCloudDataDirectoryModule(root, num_points=256, batch_size=2, num_workers=0, file_mode="voxels",
                                         output_mode="point_clouds", augmentations=False)
    dm.setup()
    assert len(dm.train_dataset) == 4 and len(dm.val_dataset) == 1
    batch = next(iter(dm.train_dataloader()))
    assert batch.shape == (2, 256, 3) and batch.dtype == torch.float32
    dv = D.PointCloudDataDirectoryModule(root, batch_size=4, num_workers=0, file_mode="voxels", output_mode="voxels",
                                         augmentations=False, relevant_object_categories=["table"])
    dv.setup()
    assert len(dv.train_dataset) + len(dv.val_dataset) == 3
    assert next(iter(dv.train_dataloader())).shape[1:] == (1, 32, 32, 32)
    with pytest.raises(RuntimeError, match="h5py"):
        D.load_sample_file(os.path.join(os.path.dirname(root), "dd", names[0][:-4] + ".dd"))
