"""Shared pieces of the two inference CLIs (counterparts of the reference's inference.py:13-38 / inference_multiview.py:13-38)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch


def preprocess_image(path, size: int = 256) -> torch.Tensor:
    """reference preprocess_image: shortest side -> `size` with PIL Lanczos, the other side int()-TRUNCATED (inference.py:19,28), centre
    crop to size x size, /255, CHW."""
    from PIL import Image

    img = Image.open(path).convert("RGB")
    W, H = img.size
    if W < H:
        new_W, new_H = size, int(H * (size / W))
        img = img.resize((new_W, new_H), Image.Resampling.LANCZOS)
        top = (new_H - size) // 2
        img = img.crop((0, top, new_W, top + size))
    else:
        new_H, new_W = size, int(W * (size / H))
        img = img.resize((new_W, new_H), Image.Resampling.LANCZOS)
        left = (new_W - size) // 2
        img = img.crop((left, 0, left + size, new_H))
    return torch.from_numpy(np.array(img).astype(np.float32)).permute(2, 0, 1) / 255.0


def normalised_intrinsics(fx, fy, cx, cy, views: int, size: int = 256) -> torch.Tensor:
    """[1, views, 3, 3]; the reference divides by its fixed 256-pixel crop (inference.py:107-115): fx, fy, cx, cy are given in pixels of
    the `size` x `size` crop."""
    K = torch.tensor([[[fx / float(size), 0, cx / float(size)], [0, fy / float(size), cy / float(size)], [0, 0, 1]]], dtype=torch.float32)
    return K.repeat(1, views, 1, 1)


def load_weights(model_path, ckpt=None):
    """state dict for SIU3RModel: a real checkpoint through siu3r_amd.checkpoint (ckpt: its already-read content), or (no path) the
    seeded synthetic weights."""
    if model_path:
        if not Path(model_path).exists():
            raise FileNotFoundError(f"Model file {model_path} does not exist.")
        from .checkpoint import load_siu3r_state_dict

        return load_siu3r_state_dict(model_path, ckpt=ckpt)
    from . import synthetic_weights as OW  # synthetic stand-in weights (plumbing run: no checkpoint is available offline)

    print("no --model_path: using seeded synthetic weights (plumbing only)", file=sys.stderr)
    return OW.make_weights(0)


def add_camera_args(ap, size_default=256):
    for k, v in dict(cx=128.0, cy=128.0, fx=318.0, fy=318.0).items():
        ap.add_argument(f"--{k}", type=float, default=None, help=f"camera intrinsic {k} in pixels of the crop (default {v} at 256, scaled with --size)")
    ap.add_argument("--size", type=int, default=size_default, help="crop size fed to the network (the reference is fixed at 256)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "bf16x3"])


def camera_from_args(a):
    s = a.size / 256.0
    d = dict(cx=128.0 * s, cy=128.0 * s, fx=318.0 * s, fy=318.0 * s)
    return [getattr(a, k) if getattr(a, k) is not None else d[k] for k in ("fx", "fy", "cx", "cy")]


def export(g, out_dir) -> Path:
    from .ply_export import export_ply

    g = g.detach_cpu_copy()
    return export_ply(g.means[0], g.scales[0], g.rotations[0], g.harmonics[0], g.opacities[0], g.semantic_labels[0], g.instance_labels[0],
                      g.seg_query_class_logits[0], Path(out_dir) / "output.ply", shift_and_scale=False, save_sh_dc_only=False)
