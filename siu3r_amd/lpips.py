"""LPIPS (VGG16) on the HIP path: the `lpips` entry of the reference evaluator's render scores (src/evaluator.py:55-57, 263, 372:
torchmetrics `LearnedPerceptualImagePatchSimilarity("vgg", normalize=True)` on every rendered target image against its ground truth).

torchmetrics (pinned 1.7.3: uv.lock:3373-3374, pyproject.toml:32; it brings its own LPIPS network, the `lpips` package is not a dependency) is a third-party dependency that is absent from /root/reference and from this image, so the
algorithm is restated from its published form (torchmetrics/functional/image/lpips.py, itself the `lpips` package's v0.1 network):
    x in [0, 1] -> 2 x - 1 -> (x - shift) / scale per channel -> torchvision VGG16 `features`, tapped behind relu1_2, relu2_2, relu3_3,
    relu4_3, relu5_3 -> per tap: unit-normalise the channel vector of every pixel (x / sqrt(1e-8 + sum_c x^2)), squared difference of the
    two images, a learned non-negative 1 x 1 convolution (one weight per channel, no bias), mean over the pixels -> sum of the five taps.
Weights: never shipped here.  The reference's Lightning checkpoints carry them (Pipeline holds the metric as a sub-module, src/pipeline.py:35:
keys `lpips.net.net.slice<k>.<i>.{weight,bias}`, `lpips.net.lin<k>.model.1.weight`, `lpips.net.scaling_layer.{shift,scale}`), and
`weights_from_state_dict` also takes a bare torchvision `features.<i>.*` + `lin<k>.model.1.weight` dict.
The convolutions run on the bf16x3 implicit-GEMM kernels (fp32 activations), pooling and the distance on two small kernels
(siu3r_maxpool2x2s2, siu3r_lpips_layer); oracle: oracle/lpips_oracle.py (parity unpinned: no torchmetrics here to generate vectors)."""
from __future__ import annotations

import re
from typing import Dict, Optional

import torch

from . import ops
from .ops import ACT_RELU

# torchvision vgg16().features: convolution indices per LPIPS slice (a max pool opens slices 2..5)
VGG_SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
VGG_CHANNELS = (64, 128, 256, 512, 512)
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
NORM_EPS = 1e-8


def weights_from_state_dict(sd: Dict[str, torch.Tensor]) -> Optional[Dict[str, torch.Tensor]]:
    """Pick the LPIPS tensors out of a state dict, whatever prefix they carry (`lpips.net.` in a Pipeline checkpoint, `net.` in the
    metric's own, none in the `lpips` package's).  Returns {"conv<i>.weight", "conv<i>.bias" (i = VGG feature index), "lin<k>" [C],
    "shift" [3], "scale" [3]} or None when the dict holds no LPIPS network."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        m = re.search(r"(?:slice\d+|features)\.(\d+)\.(weight|bias)$", k)
        if m and v.dim() in (1, 4):
            out[f"conv{int(m.group(1))}.{m.group(2)}"] = v.detach().float()
            continue
        m = re.search(r"(?:^|\.)lin(\d)\.model\.1\.weight$", k)   # (`lins.<k>.` aliases the same tensors)
        if m:
            out[f"lin{int(m.group(1))}"] = v.detach().float().reshape(-1)
            continue
        m = re.search(r"scaling_layer\.(shift|scale)$", k)
        if m:
            out[m.group(1)] = v.detach().float().reshape(-1)
    need = [f"conv{i}.{p}" for s in VGG_SLICES for i in s for p in ("weight", "bias")] + [f"lin{k}" for k in range(5)]
    if not any(k in out for k in need):
        return None
    missing = [k for k in need if k not in out]
    if missing:
        raise RuntimeError(f"LPIPS weights incomplete: missing {missing[:6]}{' ...' if len(missing) > 6 else ''}")
    out.setdefault("shift", torch.tensor(SHIFT))
    out.setdefault("scale", torch.tensor(SCALE))
    for k, c in zip(range(5), VGG_CHANNELS):
        if out[f"lin{k}"].numel() != c:
            raise RuntimeError(f"LPIPS lin{k}: {out[f'lin{k}'].numel()} weights, expected {c} (net_type 'vgg')")
    return out


class LPIPS:
    """callable (img0, img1) -> per-pair distances [N]; images [N,3,H,W] or [3,H,W] / [H,W,3] in [0, 1] (normalize=True semantics)."""

    def __init__(self, weights: Dict[str, torch.Tensor], device="cuda"):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("siu3r_amd.lpips.LPIPS runs on the GPU only (no CPU fallback)")
        w = weights_from_state_dict(weights) if not any(k.startswith("conv") for k in weights) else weights
        if w is None:
            raise RuntimeError("no LPIPS network in the given state dict")
        cin0 = ops.image_channels(True)
        self.convs = []
        for sl in VGG_SLICES:
            row = []
            for i in sl:
                wt, b = w[f"conv{i}.weight"].to(self.dev), w[f"conv{i}.bias"].to(self.dev)
                row.append(ops.pack_conv(wt, b, True, cin_pad=cin0 if wt.shape[1] == 3 else None))
            self.convs.append(row)
        self.lins = [w[f"lin{k}"].to(self.dev).contiguous() for k in range(5)]
        # 2 x - 1, then (x - shift) / scale, as ONE affine map per channel (fp32, exact up to one rounding of the constants)
        shift, scale = w["shift"].to(self.dev), w["scale"].to(self.dev)
        self.mul = (2.0 / scale).view(1, 3, 1, 1)
        self.add = ((-1.0 - shift) / scale).view(1, 3, 1, 1)

    @staticmethod
    def _nchw01(img) -> torch.Tensor:
        t = torch.as_tensor(img)
        if t.dim() == 3:
            t = t.permute(2, 0, 1) if t.shape[-1] == 3 and t.shape[0] != 3 else t
            t = t[None]
        assert t.dim() == 4 and t.shape[1] == 3, f"image batch [N,3,H,W] expected, got {tuple(t.shape)}"
        return t

    def features(self, x01: torch.Tensor):
        """[M,3,H,W] in [0,1] on the device -> the five channel-last fp32 feature maps"""
        x = (x01.to(self.dev, torch.float32) * self.mul + self.add).contiguous()   # layout / affine plumbing
        h = ops.pack_image_nhwc(x, torch.float32, ops.image_channels(True))
        taps = []
        for k, row in enumerate(self.convs):
            if k:
                h = ops.maxpool2x2s2(h)
            for pw in row:
                h = ops.conv2d(h, pw, stride=1, pad=1, out_dtype=torch.float32, act=ACT_RELU)
            taps.append(h)
        return taps

    def __call__(self, img0, img1) -> torch.Tensor:
        a, b = self._nchw01(img0), self._nchw01(img1)
        assert a.shape == b.shape and min(a.shape[2:]) >= 16, "LPIPS: equal shapes, at least 16 x 16 pixels (four 2 x 2 poolings)"
        N = a.shape[0]
        taps = self.features(torch.cat((a.to(self.dev), b.to(self.dev)), 0))
        total = torch.zeros(N, dtype=torch.float32, device=self.dev)
        for f, w in zip(taps, self.lins):
            d = ops.lpips_layer(f[:N].contiguous(), f[N:].contiguous(), w, NORM_EPS)   # [N, h, w]
            total += d.flatten(1).mean(1)
        return total
