#!/usr/bin/env python
"""Multi-view inference CLI on the MI355X path (counterpart of the reference's inference_multiview.py:41-152): every *.jpg, then *.png,
then *.jpeg of --image_dir (each group sorted, :97-101) is preprocessed like the two-view CLI, V >= 2 views go through
SIU3RMultiViewModel (view 0 -> dec_blocks / *_head1, the others -> dec_blocks2 / *_head2) and the V x H x W Gaussians are written to
output.ply (:137-150).  BASELINE.json configs[4] drives this with 8 views.

    python inference_multiview.py --image_dir assets/4views [--model_path siu3r_4view.ckpt] [--size 256]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from siu3r_amd.cli_common import add_camera_args, camera_from_args, export, load_weights, normalised_intrinsics, preprocess_image  # noqa: E402


def list_images(image_dir: Path):
    return sorted(image_dir.glob("*.jpg")) + sorted(image_dir.glob("*.png")) + sorted(image_dir.glob("*.jpeg"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_dir", default="assets/4views")
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--output_path", default="infer_outputs")
    add_camera_args(ap)
    a = ap.parse_args()
    image_dir = Path(a.image_dir)
    if not image_dir.exists():
        raise FileNotFoundError(f"Image directory {image_dir} does not exist.")
    paths = list_images(image_dir)
    if len(paths) < 2:
        raise ValueError(f"{image_dir} holds {len(paths)} image(s); at least two views are needed")
    from siu3r_amd.model import SIU3RMultiViewModel

    sd = load_weights(a.model_path)
    images = torch.stack([preprocess_image(p, a.size) for p in paths])[None]  # [1, V, 3, H, W]
    V = images.shape[1]
    fx, fy, cx, cy = camera_from_args(a)
    K = normalised_intrinsics(fx, fy, cx, cy, V, a.size)
    model = SIU3RMultiViewModel(sd, image_size=(a.size, a.size), precision=a.precision)
    with torch.no_grad():
        g, seg, masks, infos, scores = model(images.cuda(), K.cuda(), enable_query_class_logit_lift=True)
    out = export(g, a.output_path)
    print(f"wrote {out} ({V} views, {g.means.shape[1]} Gaussians, {len(infos[0])} segments)")


if __name__ == "__main__":
    main()
