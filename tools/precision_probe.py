#!/usr/bin/env python
"""CPU study of operand precision (no GPU): runs the fp32 oracle with the operands of every contraction rounded the way an
MFMA path would round them (fp32 accumulation kept), under a per-layer policy, and reports the max-normalised error of every
output tensor against the plain fp32 run.  This is how the "mixed" precision map of siu3r_amd.model was chosen.

    python tools/precision_probe.py [size] [policy ...]

policy = name:spec,name:spec...   spec in {f32, bf16, fp16, x3, fp16x2} or "act|weight" (e.g. x3|bf16); `name` is a substring of the layer (weight) name; the first
match wins; `*` matches everything; `mm` addresses the weight-free products (q k^T, p v, mask einsum).
Examples:  "*:bf16"   "*:fp16"   "downstream_head:x3,*:fp16"
"""
import os
import sys
import time

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import siu3r_oracle as O  # noqa: E402
from siu3r_amd import synthetic_weights as OW  # noqa: E402


def rnd(x, spec):
    if spec == "bf16":
        return x.to(torch.bfloat16).float()
    if spec == "fp16":
        return x.to(torch.float16).float()
    if spec == "x3":  # hi + lo split in bf16: 16 mantissa bits survive
        hi = x.to(torch.bfloat16).float()
        return hi + (x - hi).to(torch.bfloat16).float()
    if spec == "fp16x2":
        hi = x.to(torch.float16).float()
        return hi + (x - hi).to(torch.float16).float()
    return x


class Emu(TorchFunctionMode):
    def __init__(self, names, policy):
        super().__init__()
        self.names, self.policy = names, policy
        self.used = {}

    def spec(self, name):
        for pat, s in self.policy:
            if pat == "*" or pat in name:
                self.used[name] = s
                return s
        return "f32"

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.linear, F.conv2d, F.conv_transpose2d):
            x, w = args[0], args[1]
            if func is F.conv2d and kwargs.get("groups", 1) != 1:
                return func(*args, **kwargs)  # depthwise 3x3: VALU fp32 on the GPU path
            s = self.spec(self.names.get(id(w), "?unnamed"))
            if s != "f32":
                sa, sw = s.split("|") if "|" in s else (s, s)  # "x3|bf16": activations hi+lo, weights one bf16 plane (a 2-pass product)
                args = (rnd(x, sa), rnd(w, sw)) + tuple(args[2:])
            return func(*args, **kwargs)
        if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.bmm):
            s = self.spec("mm")
            if s != "f32":
                sa, sb = s.split("|") if "|" in s else (s, s)  # "x3|bf16": first operand (q, p) hi+lo, second (k^T, v) one bf16 plane
                args = (rnd(args[0], sa), rnd(args[1], sb))
            return func(*args, **kwargs)
        if func is torch.einsum:
            s = self.spec("mm.einsum")
            if s != "f32":
                args = (args[0],) + tuple(rnd(a, s) for a in args[1:])
            return func(*args, **kwargs)
        return func(*args, **kwargs)


FIELDS = ("means", "covariances", "harmonics", "opacities", "scales", "rotations", "class_queries_logits", "masks_queries_logits")


def run(sd, names, img, K, policy):
    if policy is None:
        with torch.no_grad():
            return O.model_forward(sd, img, K, keep_intermediates=False), {}
    m = Emu(names, policy)
    with torch.no_grad(), m:
        out = O.model_forward(sd, img, K, keep_intermediates=False)
    return out, m.used


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    pols = sys.argv[2:] or ["*:bf16", "*:fp16"]
    torch.set_num_threads(8)
    sd = OW.make_weights(0)
    names = {id(v): k for k, v in sd.items()}
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 2, 3, size, size, generator=g)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1)
    t0 = time.time()
    ref, _ = run(sd, names, img, K, None)
    print(f"fp32 reference {time.time() - t0:.1f} s; segments {ref['seg_infos']}")
    for pol in pols:
        policy = [tuple(p.split(":")) for p in pol.split(",")]
        out, used = run(sd, names, img, K, policy)
        es = {f: float((out[f] - ref[f]).abs().max() / (ref[f].abs().max() + 1e-30)) for f in FIELDS}
        lab = bool(torch.equal(out["semantic_labels"], ref["semantic_labels"]))
        n16 = sum(1 for v in used.values() if v in ("bf16", "fp16"))
        print(f"{pol:60s} worst {max(es.values()):.2e} | " + " ".join(f"{k[:5]}={v:.1e}" for k, v in es.items()) + f" | labels_eq={lab} | 16-bit layers {n16}/{len(used)}")


if __name__ == "__main__":
    main()
