"""Dataset front-end (customnerf_amd/nerf/provider.py): pose normalisation against the reference's own
auto_orient_and_center_poses (golden, tests/golden/orient.npz), transforms.json parsing, focal lengths; the GPU part builds a small
scene on disk and checks rays against the oracle's _generate_rays restatement."""
import json
import os

import numpy as np
import pytest
import torch


def test_pose_normalisation_matches_reference_golden(golden):
    from customnerf_amd.nerf.provider import auto_orient_and_center_poses
    g = golden("orient")
    for method in ("up", "none"):
        for center in (1, 0):
            out, tr = auto_orient_and_center_poses(torch.from_numpy(g["poses"]), method=method, center_poses=bool(center))
            np.testing.assert_allclose(out.numpy(), g[f"{method}_{center}__poses"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(tr.numpy(), g[f"{method}_{center}__transform"], rtol=0, atol=2e-6)


def test_focal_lengths():
    from customnerf_amd.nerf.provider import focal_lengths
    assert focal_lengths({"fl_x": 500.0, "fl_y": 510.0}) == (500.0, 510.0)
    fx, fy = focal_lengths({"camera_angle_x": 1.0, "camera_angle_y": 0.8, "w": 640, "h": 480})
    assert abs(fx - 0.5 * 640 / np.tan(0.5)) < 1e-4 and abs(fy - 0.5 * 480 / np.tan(0.4)) < 1e-4
    fx, _ = focal_lengths({"x_fov": 60.0, "y_fov": 45.0, "w": 100, "h": 80})
    assert abs(fx - 0.5 * 100 / np.tan(np.deg2rad(30.0))) < 1e-4
    with pytest.raises(AttributeError):
        focal_lengths({"w": 10, "h": 10})


def _write_scene(root, V=10, H=24, W=32):
    from PIL import Image
    os.makedirs(os.path.join(root, "images"))
    os.makedirs(os.path.join(root, "masks"))
    rng = np.random.default_rng(0)
    frames = []
    for i in range(V):
        th = 2 * np.pi * i / V
        eye = np.array([3 * np.cos(th), 3 * np.sin(th), 1.0])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, eye
        name = f"images/frame_{V - i:03d}.jpg"                       # reversed names: the loader sorts by file_path
        frames.append({"file_path": name, "transform_matrix": m.tolist()})
        Image.fromarray(rng.integers(0, 255, (H * 2, W * 2, 3), dtype=np.uint8)).save(os.path.join(root, name), quality=95)
        if i % 2 == 0:
            mk = np.zeros((H * 2, W * 2), np.uint8); mk[H // 2:H, W // 2:W] = 255
            Image.fromarray(mk).save(os.path.join(root, name.replace("images", "masks").replace(".jpg", ".png")))
    json.dump({"fl_x": 60.0, "fl_y": 60.0, "cx": W * 1.0, "cy": H * 1.0, "w": W * 2, "h": H * 2, "frames": frames}, open(os.path.join(root, "transforms.json"), "w"))


def test_scene_host_side_without_gpu(tmp_path, monkeypatch):
    """parsing, ordering, train split, pose scaling (no images decoded onto a device, no rays)"""
    from customnerf_amd.nerf import provider
    _write_scene(str(tmp_path))
    monkeypatch.setattr("customnerf_amd.nerf.provider_utils.generate_rays", lambda c2w, *a, **k: (torch.zeros(c2w.shape[0], a[4], a[5], 3), torch.zeros(c2w.shape[0], a[4], a[5], 3)))
    sc = provider.NerfstudioScene(str(tmp_path), resolution_level=2, device="cpu")
    assert len(sc) == 9 and sc.H == 24 and sc.W == 32                         # ceil(0.9 * 10) views, area-resized by 2
    assert sc.image_paths[0].endswith("frame_001.jpg") and sc.image_paths == sorted(sc.image_paths)
    assert abs(float(sc.camera_to_world[:, :3, 3].abs().max()) - 1.0) < 0.35  # scaled by the max |t| over ALL frames (:233-236)
    assert sc.images.shape == (9, 24, 32, 3) and float(sc.images.max()) < 1.0 and sc.masks.shape == (9, 24, 32)
    assert set(np.unique(sc.masks.numpy())) <= {0.0, 1.0} and float(sc.masks.sum()) > 0
    rgbs, mask, ro, rd, H, W, path = sc[3]
    assert rgbs.shape == (1, 24 * 32, 3) and mask.shape == (1, 24 * 32) and (H, W) == (24, 32) and path == sc.image_paths[3]


@pytest.mark.gpu
def test_scene_rays_match_oracle(tmp_path):
    from customnerf_amd.nerf import provider
    from oracle import torch_oracle as to
    _write_scene(str(tmp_path))
    sc = provider.NerfstudioScene(str(tmp_path), resolution_level=2, device="cuda")
    o_ref, d_ref = to.generate_rays(sc.camera_to_world, sc.fx, sc.fy, sc.cx, sc.cy, sc.H, sc.W, level=2.0)
    np.testing.assert_allclose(sc.rays_d.view(len(sc), sc.H, sc.W, 3).cpu().numpy(), d_ref.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(sc.rays_o.view(len(sc), sc.H, sc.W, 3).cpu().numpy(), o_ref.numpy(), rtol=0, atol=1e-6)
