#!/usr/bin/env python
"""Two-image inference CLI on the MI355X path (counterpart of the reference's inference.py:41-150): PIL Lanczos resize
of the shorter side + centre crop (:13-38), normalised intrinsics (:107-115), SIU3RModel forward, output.ply (:137-150).

    python inference.py --image_path1 a.jpg --image_path2 b.jpg [--model_path siu3r_epoch100.ckpt] [--size 256]

Without --model_path the seeded synthetic weights are used (no checkpoint is available offline): the output then only
exercises the plumbing (BASELINE.json configs[0])."""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from siu3r_amd.cli_common import add_camera_args, camera_from_args, export, load_weights, normalised_intrinsics, preprocess_image  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_path1", required=True)
    ap.add_argument("--image_path2", required=True)
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--output_path", default="outputs")
    add_camera_args(ap)
    a = ap.parse_args()
    for p in (a.image_path1, a.image_path2):
        if not Path(p).exists():
            raise FileNotFoundError(f"Image file {p} does not exist.")
    from siu3r_amd.model import SIU3RModel

    sd = load_weights(a.model_path)
    images = torch.stack([preprocess_image(a.image_path1, a.size), preprocess_image(a.image_path2, a.size)])[None]
    fx, fy, cx, cy = camera_from_args(a)
    K = normalised_intrinsics(fx, fy, cx, cy, 2, a.size)
    model = SIU3RModel(sd, image_size=(a.size, a.size), precision=a.precision)
    with torch.no_grad():
        g, seg, masks, infos, scores = model(images.cuda(), K.cuda(), enable_query_class_logit_lift=True)
    out = export(g, a.output_path)
    print(f"wrote {out} ({g.means.shape[1]} Gaussians, {len(infos[0])} segments)")


if __name__ == "__main__":
    main()
