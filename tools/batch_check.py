"""Per-item differences between a batch-of-B forward and the same pairs run alone (bf16x3, 512^2).  python tools/batch_check.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from golden_utils import fixture_images, default_K
from siu3r_amd import synthetic_weights as OW
from siu3r_amd.model import SIU3RModel
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 512
g = torch.Generator().manual_seed(11)
fx_ = fixture_images(S)
img = torch.cat([fx_, torch.rand(B - 2, 2, 3, S, S, generator=g), fx_.flip(1)]).cuda()
K = default_K().repeat(B, 1, 1, 1).cuda()
m = SIU3RModel(OW.make_weights(0), image_size=(S, S), precision="bf16x3")
rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
with torch.no_grad():
    outs = [m(img, K, enable_query_class_logit_lift=True) for _ in range(3)]  # eager, capture, replay: all three kept alive
    gs, seg, masks, infos, qs = outs[2]
    if os.environ.get("BC_TOUCH"):
        print("touch", float(masks[0].float().sum()), float(gs.seg_query_class_logits[0].float().sum()), gs.semantic_labels[0:1].cpu().sum().item())
    for i in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else range(B)):
        one = m(img[i:i + 1], K[i:i + 1], enable_query_class_logit_lift=True)
        print(i, "means %.1e cov %.1e sh %.1e op %.1e | class %.1e mask %.1e | sem agree %.5f ins agree %.5f | segs %d vs %d" % (
            rel(gs.means[i], one[0].means[0]), rel(gs.covariances[i], one[0].covariances[0]), rel(gs.harmonics[i], one[0].harmonics[0]), rel(gs.opacities[i], one[0].opacities[0]),
            rel(seg.class_queries_logits[i], one[1].class_queries_logits[0]), rel(seg.masks_queries_logits[i], one[1].masks_queries_logits[0]),
            float((gs.semantic_labels[i] == one[0].semantic_labels[0]).float().mean()), float((gs.instance_labels[i] == one[0].instance_labels[0]).float().mean()),
            len(infos[i]), len(one[3][0])))
