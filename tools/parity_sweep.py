"""Shape / seed sweep of the HIP forward (bf16x3) against the CPU oracle: odd aspect ratios, small sizes, B = 2, V = 3 and 4.
python tools/parity_sweep.py  (needs a GPU; ~1 min)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import siu3r_oracle as O, weights as OW
from siu3r_amd.model import SIU3RModel, SIU3RMultiViewModel

torch.set_num_threads(16)
sd = OW.make_weights(0)
def err(a, b): return float((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
worst = 0.0
cases = [(1, 2, 64, 64, 1), (1, 2, 96, 160, 2), (1, 2, 160, 96, 3), (2, 2, 64, 96, 4), (1, 3, 96, 96, 5), (1, 4, 64, 64, 6), (1, 2, 224, 224, 7)]
for (B, V, H, W, seed) in cases:
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, V, 3, H, W, generator=g)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, V, 1, 1)
    K[:, :, 0, 0] *= 1.0 + 0.1 * torch.rand(B, V, generator=g)
    cls = SIU3RModel if V == 2 else SIU3RMultiViewModel
    model = cls(sd, image_size=(H, W), precision="bf16x3")
    with torch.no_grad():
        ref = (O.model_forward if V == 2 else O.model_forward_multi)(sd, img, K, keep_intermediates=False)
        outs = [model(img.cuda(), K.cuda()) for _ in range(3)]   # eager, capture, replay
    gs, seg = outs[2][0], outs[2][1]
    es = {f: err(getattr(gs, f), ref[f]) for f in ("means", "covariances", "harmonics", "opacities", "scales", "rotations")}
    es["class"] = err(seg.class_queries_logits, ref["class_queries_logits"])
    es["mask"] = err(seg.masks_queries_logits, ref["masks_queries_logits"])
    agree = min(float((gs.semantic_labels.cpu() == ref["semantic_labels"]).float().mean()), float((gs.instance_labels.cpu() == ref["instance_labels"]).float().mean()))
    ok_int = agree >= 0.995  # non-empty panoptic result: border pixels may change owner (argmax over fp32 scores)
    same = torch.equal(outs[0][0].means, outs[2][0].means)
    w = max(es.values()); worst = max(worst, w)
    print(f"B={B} V={V} {H}x{W}: worst {w:.2e} ({max(es, key=es.get)}) segments={[len(i) for i in outs[2][3]]} labels_agree={agree:.5f} replay_identical={same}")
    if B > 1 and V == 2:  # per-item errors and the same items run alone: batch-dependent behaviour would show here
        for i in range(B):
            with torch.no_grad():
                one = model(img[i:i + 1].cuda(), K[i:i + 1].cuda())
            r_i = ref["masks_queries_logits"][i:i + 1]
            print(f"   item {i}: mask err in batch {err(seg.masks_queries_logits[i:i + 1], r_i):.2e}, alone {err(one[1].masks_queries_logits, r_i):.2e}, batch-vs-alone {err(seg.masks_queries_logits[i:i + 1], one[1].masks_queries_logits.float().cpu()):.2e}")
    if w > 1e-3:
        print("   all errors:", {k: f"{v:.1e}" for k, v in es.items()})
    # Gaussian fields have no discrete dependence: 1e-3 always.  The Mask2Former logits pass through nine thresholded attention
    # masks (sigmoid < 0.5): a borderline pixel can flip between two fp32 evaluation orders, which moves those logits by a few 1e-3
    assert max(v for k, v in es.items() if k not in ("class", "mask")) <= 1e-3 and w <= 1e-2 and ok_int and same
    del model; torch.cuda.empty_cache()
print("sweep ok, worst", worst)
