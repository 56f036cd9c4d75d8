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

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def preprocess_image(path, size=256):
    from PIL import Image

    img = Image.open(path).convert("RGB")
    w, h = img.size
    s = size / min(w, h)
    img = img.resize((max(size, round(w * s)), max(size, round(h * s))), Image.LANCZOS)
    w, h = img.size
    l, t = (w - size) // 2, (h - size) // 2
    img = img.crop((l, t, l + size, t + size))
    return torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_path1", required=True)
    ap.add_argument("--image_path2", required=True)
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--output_path", default="outputs")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "bf16x3"])
    for k, v in dict(cx=128.0, cy=128.0, fx=318.0, fy=318.0).items():
        ap.add_argument(f"--{k}", type=float, default=v)
    a = ap.parse_args()
    for p in (a.image_path1, a.image_path2):
        if not Path(p).exists():
            raise FileNotFoundError(f"Image file {p} does not exist.")
    from siu3r_amd.model import SIU3RModel
    from siu3r_amd.ply_export import export_ply

    if a.model_path:
        if not Path(a.model_path).exists():
            raise FileNotFoundError(f"Model file {a.model_path} does not exist.")
        ck = torch.load(a.model_path, map_location="cpu", weights_only=False)
        sd = ck.get("state_dict", ck.get("model", ck))
    else:
        from siu3r_amd import synthetic_weights as OW  # synthetic stand-in weights (plumbing run)

        print("no --model_path: using seeded synthetic weights (plumbing only)", file=sys.stderr)
        sd = OW.make_weights(0)
    images = torch.stack([preprocess_image(a.image_path1, a.size), preprocess_image(a.image_path2, a.size)])[None]
    K = torch.tensor([[[a.fx / 256.0, 0, a.cx / 256.0], [0, a.fy / 256.0, a.cy / 256.0], [0, 0, 1]]]).repeat(1, 2, 1, 1)
    model = SIU3RModel(sd, image_size=(a.size, a.size), precision=a.precision)
    with torch.no_grad():
        g, seg, masks, infos, scores = model(images.cuda(), K.cuda(), enable_query_class_logit_lift=True)
    g = g.detach_cpu_copy()
    out = export_ply(g.means[0], g.scales[0], g.rotations[0], g.harmonics[0], g.opacities[0], g.semantic_labels[0], g.instance_labels[0],
                     g.seg_query_class_logits[0], Path(a.output_path) / "output.ply", shift_and_scale=False, save_sh_dc_only=False)
    print(f"wrote {out} ({g.means.shape[1]} Gaussians, {len(infos[0])} segments)")


if __name__ == "__main__":
    main()
