#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference, imported through tests/golden/_ref_import.py with stub
modules for the packages the image lacks).  Nothing of the reference is copied: the fixtures hold inputs' seeds and
OUTPUT values only (strided samples + norms for the 100+ MB tensors).  Weights are regenerated on both sides from
oracle/weights.py (build-owned counter-based generator), never stored.

    python tests/golden/make_golden.py            # writes *.npz / *.json next to this file
"""
import json
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import _ref_import as R  # noqa: E402
from crafted import crafted_panoptic_inputs, permuted  # noqa: E402
from oracle import weights as OW  # noqa: E402

STRIDE = 4093  # prime stride for the sampled tensors
ISTRIDE = 97    # stride for the integer id maps
WINDOW = 65536  # one contiguous window per tensor, compared densely (offset: a third into the tensor, 64-aligned)


def window_of(numel):
    if numel <= WINDOW:
        return 0, numel
    return (numel // 3) // 64 * 64, WINDOW


def summarize(t: torch.Tensor):
    f = t.detach().float().reshape(-1)
    o, n = window_of(f.numel())
    return dict(shape=list(t.shape), l2=float(f.double().norm()), mean=float(f.double().mean()), absmax=float(f.abs().max()),
                sample=f[::STRIDE].numpy().astype(np.float32), window=f[o:o + n].numpy().astype(np.float32))


def load_pair(size):
    from PIL import Image

    imgs = []
    for i in (1, 2):
        im = Image.open(f"/root/reference/assets/living_room_image{i}.jpg").convert("RGB")
        a = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255.0
        if a.shape[-1] != size:
            a = torch.nn.functional.interpolate(a[None], size=(size, size), mode="bilinear", align_corners=False)[0]
        imgs.append(a)
    return torch.stack(imgs)[None]


def model_fixture(size, sd):
    model = R.build_reference_model((size, size))
    model.load_state_dict(sd, strict=True)
    img = load_pair(size)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1)
    with torch.no_grad():
        g, seg, masks, infos, qs = model(img, K, enable_query_class_logit_lift=True)
    out = {}
    for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        for k, v in summarize(getattr(g, f)).items():
            out[f"{f}.{k}"] = np.asarray(v)
    for name, t in (("class_queries_logits", seg.class_queries_logits), ("masks_queries_logits", seg.masks_queries_logits)):
        for k, v in summarize(t).items():
            out[f"{name}.{k}"] = np.asarray(v)
    out["class_queries_logits.full"] = seg.class_queries_logits.numpy()
    out["semantic_labels.sum"] = np.asarray(int(g.semantic_labels.sum()))
    out["instance_labels.sum"] = np.asarray(int(g.instance_labels.sum()))
    out["seg_mask.dtype"] = np.asarray(str(masks[0].dtype))
    out["seg_mask.unique"] = masks[0].unique().numpy()
    # integer outputs of the (non-empty) panoptic branch: strided samples + histograms of the id maps, samples of the query x class
    # logit volume the lifting consumes
    out["seg_mask.sample"] = masks[0].reshape(-1)[::ISTRIDE].numpy()
    out["seg_mask.hist"] = torch.bincount(masks[0].reshape(-1).long().clamp_min(0), minlength=8).numpy()
    out["semantic_labels.sample"] = g.semantic_labels.reshape(-1)[::ISTRIDE].numpy()
    out["instance_labels.sample"] = g.instance_labels.reshape(-1)[::ISTRIDE].numpy()
    out["semantic_labels.hist"] = torch.bincount(g.semantic_labels.reshape(-1).long(), minlength=22).numpy()
    qcl = g.seg_query_class_logits[0]
    out["qcl.shape"] = np.asarray(qcl.shape)
    out["qcl.sample"] = qcl.reshape(-1)[::STRIDE].numpy()
    out["qcl.l2"] = np.asarray(float(qcl.double().norm()))
    out["image.small"] = torch.nn.functional.interpolate(img[0], size=(32, 32), mode="area").numpy()  # input checksum aid
    np.savez_compressed(os.path.join(HERE, f"model_{size}.npz"), **out)
    with open(os.path.join(HERE, f"model_{size}.json"), "w") as fh:
        json.dump(dict(seg_infos=infos, query_scores=qs, input="assets/living_room_image{1,2}.jpg, /255, bilinear to size",
                       intrinsics="fx=fy=318/256, c=0.5", weights="oracle.weights.make_weights(0)", stride=STRIDE), fh)
    print("model fixture", size, "done")


def multi_fixture(sd, size=128, V=3):
    """SIU3RMultiViewModel (src/models/model_multi.py) on V views made of the asset pair (golden_utils.multi_views: V = 3 is the pair + the
    first image mirrored; V = 8 adds the other mirrors and the transposes).  Fields and logits as strided samples + dense windows + norms;
    the integer outputs of the panoptic branch as strided samples + whole-map histograms, like the two-view fixtures."""
    from golden_utils import multi_views

    model = R.build_reference_model((size, size), multi=True)
    model.load_state_dict(sd, strict=False)  # backbone.mask_token (unused at inference) is not in the key spec
    img = multi_views(load_pair(size)[0], V)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, V, 1, 1)
    with torch.no_grad():
        g, seg, masks, infos, qs = model(img, K, enable_query_class_logit_lift=True)
    out = {}
    for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations"):
        for k, v in summarize(getattr(g, f)).items():
            out[f"{f}.{k}"] = np.asarray(v)
    for name, t in (("class_queries_logits", seg.class_queries_logits), ("masks_queries_logits", seg.masks_queries_logits)):
        for k, v in summarize(t).items():
            out[f"{name}.{k}"] = np.asarray(v)
    out["class_queries_logits.full"] = seg.class_queries_logits.numpy()
    out["semantic_labels.sum"] = np.asarray(int(g.semantic_labels.sum()))
    out["instance_labels.sum"] = np.asarray(int(g.instance_labels.sum()))
    out["seg_mask.dtype"] = np.asarray(str(masks[0].dtype))
    out["seg_mask.sample"] = masks[0].reshape(-1)[::ISTRIDE].numpy()
    out["seg_mask.hist"] = torch.bincount(masks[0].reshape(-1).long().clamp_min(0), minlength=8).numpy()
    out["semantic_labels.sample"] = g.semantic_labels.reshape(-1)[::ISTRIDE].numpy()
    out["instance_labels.sample"] = g.instance_labels.reshape(-1)[::ISTRIDE].numpy()
    out["semantic_labels.hist"] = torch.bincount(g.semantic_labels.reshape(-1).long(), minlength=22).numpy()
    qcl = g.seg_query_class_logits[0]
    out["qcl.shape"] = np.asarray(qcl.shape)
    out["qcl.sample"] = qcl.reshape(-1)[::STRIDE].numpy()
    out["qcl.l2"] = np.asarray(float(qcl.double().norm()))
    np.savez_compressed(os.path.join(HERE, f"model_multi_v{V}_{size}.npz"), **out)
    with open(os.path.join(HERE, f"model_multi_v{V}_{size}.json"), "w") as fh:
        json.dump(dict(seg_infos=infos, query_scores=qs, input="golden_utils.multi_views of the asset pair, /255, bilinear to size", views=V,
                       intrinsics="fx=fy=318/256, c=0.5", weights="oracle.weights.make_weights(0)", stride=STRIDE), fh)
    print("multi-view fixture done", V, size, [len(i) for i in infos])


def panoptic_fixture():
    from src.models.mask2former.image_processing_video_mask2former import VideoMask2FormerImageProcessor
    from src.models.mask2former.video_seg_decoder import VideoMask2FormerForVideoSegmentationOutput as Out

    store, meta = {}, {}
    for oname, order in (("natural", [0, 1, 2, 3]), ("quirk_first", [2, 0, 3, 1])):
        cls, msk = permuted(*crafted_panoptic_inputs(), order)
        ref = VideoMask2FormerImageProcessor().post_process_panoptic_segmentation(
            Out(class_queries_logits=cls, masks_queries_logits=msk), threshold=0.5, target_sizes=[(64, 64)] * 4, label_ids_to_fuse={0, 1})
        meta[oname] = []
        for b, r in enumerate(ref):
            store[f"{oname}.{b}.segmentation"] = r["segmentation"].numpy()
            store[f"{oname}.{b}.qcl_shape"] = np.asarray(r["query_class_logits"].shape)
            store[f"{oname}.{b}.qcl_sample"] = r["query_class_logits"].reshape(-1)[::97].numpy()
            meta[oname].append(dict(segments_info=r["segments_info"], query_scores=r["query_scores"], dtype=str(r["segmentation"].dtype)))
    np.savez_compressed(os.path.join(HERE, "panoptic_crafted.npz"), **store)
    with open(os.path.join(HERE, "panoptic_crafted.json"), "w") as fh:
        json.dump(meta, fh)
    print("panoptic fixture done")


def lifting_fixture():
    """Executes the reference's own lifting statements (src/pipeline.py, the body of the per-item loop in
    step_w_query_class_logit_lift) on synthetic rendered logits.  The source text is read and exec'd at generation
    time only; it is not stored."""
    src = open("/root/reference/src/pipeline.py").read().split("\n")
    start = next(i for i, l in enumerate(src) if "v, q, c, h, w = render_qc_logit.shape" in l)
    end = next(i for i, l in enumerate(src) if "all_sem_id.append(sem_id)" in l)
    body = "\n".join(l[12:] if l.startswith(" " * 12) else l.lstrip() for l in src[start:end])
    gen = torch.Generator().manual_seed(21)
    cases = {}
    for name, (v, q, c, h, w) in dict(a=(2, 3, 21, 24, 32), b=(3, 5, 21, 16, 16)).items():
        x = torch.rand(v, q, c, h, w, generator=gen) * torch.rand(v, q, 1, h, w, generator=gen)
        x[:, :, -1] *= 0.5
        x[0, 0, 1, :4] = 5.0  # wall stuff region owned by query 1
        scores = [round(0.9 - 0.1 * i, 6) for i in range(q)]
        cfg = types.SimpleNamespace(model=types.SimpleNamespace(mask2former=types.SimpleNamespace(label_ids_to_fuse=[0, 1], num_queries=100)))
        env = dict(torch=torch, render_qc_logit=x.clone(), q_score=scores, self=types.SimpleNamespace(device="cpu", pipecfg=cfg))
        exec(body, env)
        cases[name] = dict(shape=[v, q, c, h, w], scores=scores, info=env["info"])
        np.savez_compressed(os.path.join(HERE, f"lifting_{name}.npz"), x=x.numpy(), sem_id=env["sem_id"].numpy(), ins_id=env["q_index"].numpy())
    with open(os.path.join(HERE, "lifting.json"), "w") as fh:
        json.dump(cases, fh)
    print("lifting fixture done")


def small_op_fixtures():
    from src.models.croco.pos_embed import RoPE2D  # PyTorch fallback of the reference
    from src.utils.projection import get_fov

    gen = torch.Generator().manual_seed(5)
    tok = torch.rand(2, 4, 33, 64, generator=gen) * 2 - 1
    pos = torch.randint(0, 33, (2, 33, 2), generator=gen)
    rope = RoPE2D(freq=100.0)(tok.clone(), pos)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None]
    fov = get_fov(K)
    # get_projection_matrix lives in a module that imports the CUDA rasterizer: stub that import
    for m in ("diff_gaussian_rasterization",):
        mod = types.ModuleType(m)
        mod.GaussianRasterizationSettings = mod.GaussianRasterizer = object
        sys.modules[m] = mod
    from src.models.cuda_splatting import get_projection_matrix

    proj = get_projection_matrix(torch.tensor([1.0]), torch.tensor([1000.0]), fov[:, 0], fov[:, 1])
    from src.models.gaussian_adapter import UnifiedGaussianAdapter

    raw = (torch.rand(3, 50, 83, generator=gen) * 2 - 1) * 4
    ga = UnifiedGaussianAdapter(0.5, 15.0, 4).forward(torch.zeros(3, 50, 3), raw)
    np.savez_compressed(os.path.join(HERE, "small_ops.npz"), rope_tok=tok.numpy(), rope_pos=pos.numpy(), rope_out=rope.numpy(),
                        fov=fov.numpy(), proj=proj.numpy(), ga_raw=raw.numpy(), ga_cov=ga.covariances.numpy(), ga_sh=ga.harmonics.numpy(),
                        ga_op=ga.opacities.numpy(), ga_scale=ga.scales.numpy(), ga_rot=ga.rotations.numpy())
    print("small-op fixtures done")


if __name__ == "__main__":
    assert R.reference_available(), "/root/reference is required to (re)generate the fixtures"
    R.install_stubs()
    torch.manual_seed(0)
    if "models" not in sys.argv[1:]:  # `make_golden.py models`: only the forward fixtures
        small_op_fixtures()
        panoptic_fixture()
        lifting_fixture()
    sd = OW.make_weights(0)
    multi_fixture(sd)
    multi_fixture(sd, size=256, V=8)   # BASELINE configs[4]'s network half (8 views), at 256^2: 20 s of reference CPU time
    if "multi" in sys.argv[1:]:        # `make_golden.py models multi`: only the multi-view fixtures
        sys.exit(0)
    for size in (256, 512):
        model_fixture(size, sd)
