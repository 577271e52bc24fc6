"""Dataset front-end for nerfstudio-format scenes: counterpart of NerfstudioData._load_renderings /
get_focal_lengths / _generate_rays (nerf/provider.py:201-470) + auto_orient_and_center_poses (nerf/provider_utils.py:35-115),
feeding the ray kernel (cnerf_generate_rays).  Host-side plumbing: JSON, PIL decoding and a handful of 4x4 matrices on the CPU;
rays, images and masks end up resident on the GPU, one entry per training view.

Differences from the reference, by design: images are area-resized with PIL's BOX filter (cv2 is not a dependency; identical to
cv2.INTER_AREA for integer `resolution_level`), everything stays on the device (the reference keeps numpy lists and re-uploads),
the OPENCV_FISHEYE undistortion branch is not built (pinhole only; raises)."""
import json
import math
import os

import numpy as np
import torch


def rotation_matrix(a, b):
    """provider_utils.py:35-57: rotation taking direction a to direction b (Rodrigues)."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    c = torch.dot(a, b)
    if c < -1 + 1e-8:                                           # exactly opposite: perturb (the reference draws torch.rand here)
        eps = (torch.rand(3) - 0.5) * 0.01
        return rotation_matrix(a + eps, b)
    s = torch.linalg.norm(v)
    k = torch.tensor([[0.0, -float(v[2]), float(v[1])], [float(v[2]), 0.0, -float(v[0])], [-float(v[1]), float(v[0]), 0.0]])
    return torch.eye(3) + k + k @ k * ((1 - c) / (s ** 2 + 1e-8))


def auto_orient_and_center_poses(poses, method="up", center_poses=True):
    """provider_utils.py:60-115 ("up" and "none"; "pca" is never used by the reference's loader).  poses [V,4,4] ->
    (oriented [V,3,4], transform [3,4])."""
    poses = torch.as_tensor(poses, dtype=torch.float32)
    translation = poses[..., :3, 3]
    mean_translation = torch.mean(translation, dim=0)
    translation = mean_translation if center_poses else torch.zeros_like(mean_translation)
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
    elif method == "none":
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
    else:
        raise ValueError(f"unsupported orientation method {method!r}")
    return transform @ poses, transform


def focal_lengths(meta):
    """provider.py:296-322"""
    def f(rad, res):
        return 0.5 * res / np.tan(0.5 * rad)
    fl_x = meta["fl_x"] if "fl_x" in meta else f(np.deg2rad(meta["x_fov"]), meta["w"]) if "x_fov" in meta else \
        f(meta["camera_angle_x"], meta["w"]) if "camera_angle_x" in meta else 0
    fl_y = meta["fl_y"] if "fl_y" in meta else f(np.deg2rad(meta["y_fov"]), meta["h"]) if "y_fov" in meta else \
        f(meta["camera_angle_y"], meta["h"]) if "camera_angle_y" in meta else 0
    if fl_x == 0 or fl_y == 0:
        raise AttributeError("Focal length cannot be calculated from transforms.json (missing fields).")
    return float(fl_x), float(fl_y)


def _load_image(path, level, mode):
    from PIL import Image
    img = Image.open(path).convert(mode)
    if level != 1:
        img = img.resize((int(img.width / level), int(img.height / level)), Image.BOX)
    return np.asarray(img, dtype=np.float32) / 256.0          # the reference divides by 256 (provider.py:268)


class NerfstudioScene:
    """One nerfstudio-format scene resident on the GPU.  scene[i] -> (rgbs [1,N,3], mask [1,N], rays_o [1,N,3], rays_d [1,N,3], H, W,
    img_path): the tuple the reconstruction / editing steps take (utils_init_nerf.py:195, 354)."""

    def __init__(self, data_dir, keyword="masks", resolution_level=1, device="cuda", train_fraction=0.9, load_images=True):
        from .provider_utils import generate_rays
        self.data_dir, self.level = data_dir, resolution_level
        jf = os.path.join(data_dir, "transforms.json")
        if not os.path.exists(jf):
            jf = os.path.join(data_dir, "transforms_train.json")
        with open(jf, encoding="UTF-8") as fh:
            meta = json.load(fh)
        # provider.py:254 + :354-359: OPENCV_FISHEYE scenes go through the un-distortion branch of the ray generator
        self.distortion = ([float(meta.get(k, 0.0)) for k in ("k1", "k2", "k3", "k4", "p1", "p2")]
                           if meta.get("camera_model") == "OPENCV_FISHEYE" else None)
        self.meta = meta
        frames = sorted(meta["frames"], key=lambda x: x["file_path"])                                  # provider.py:218
        poses = np.array([np.array(f["transform_matrix"]) for f in frames]).astype(np.float32)
        images_lis = [os.path.join(data_dir, f["file_path"]) for f in frames]
        masks_lis = [t.replace("images", keyword).replace(".jpg", ".png").replace(".JPG", ".png") for t in images_lis]
        poses, self.transform = auto_orient_and_center_poses(torch.tensor(poses), method="up", center_poses=True)
        poses = poses.numpy()
        poses[:, :3, 3] *= 1.0 / float(np.max(np.abs(poses[:, :3, 3])))                               # :233-236
        n = len(images_lis)
        i_train = np.linspace(0, n - 1, math.ceil(n * train_fraction), dtype=int)                      # :243-248
        self.image_paths = [images_lis[i] for i in i_train]
        self.mask_paths = [masks_lis[i] for i in i_train]
        self.camera_to_world = torch.from_numpy(poses[i_train][:, :3]).contiguous()
        self.n_images = len(self.image_paths)
        self.fx, self.fy = focal_lengths(meta)
        self.cx, self.cy = float(meta["cx"]), float(meta["cy"])
        if load_images:
            imgs = [_load_image(p, resolution_level, "RGB") for p in self.image_paths]
            self.H, self.W = imgs[0].shape[0], imgs[0].shape[1]
            masks = []
            for p in self.mask_paths:
                if not os.path.isfile(p):
                    masks.append(np.zeros((self.H, self.W), np.float32))                               # :279-282
                    continue
                from PIL import Image
                m = Image.open(p).convert("L").resize((self.W, self.H), Image.BOX)
                m = np.asarray(m, dtype=np.float32) / 256.0
                masks.append((m > 0).astype(np.float32))                                               # mask[mask > 0] = True
            self.images = torch.from_numpy(np.stack(imgs)).to(device)                                  # [V,H,W,3]
            self.masks = torch.from_numpy(np.stack(masks)).to(device)                                  # [V,H,W]
        else:
            self.H, self.W = int(meta["h"] / resolution_level), int(meta["w"] / resolution_level)
            self.images = self.masks = None
        o, d = generate_rays(self.camera_to_world.to(device), self.fx, self.fy, self.cx, self.cy, self.H, self.W, float(resolution_level), "nerfstudio",
                             distortion=self.distortion)
        self.rays_o, self.rays_d = o.view(self.n_images, 1, -1, 3), d.view(self.n_images, 1, -1, 3)

    def __len__(self):
        return self.n_images

    def __getitem__(self, i):
        rgbs = self.images[i].reshape(1, -1, 3) if self.images is not None else None
        mask = self.masks[i].reshape(1, -1) if self.masks is not None else None
        return rgbs, mask, self.rays_o[i], self.rays_d[i], self.H, self.W, self.image_paths[i]
