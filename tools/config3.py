"""BASELINE configs[2] shape on synthetic data (SURVEY.md 8(d) "Config 3"): B = 8 pairs @512^2, 6 target views per pair, the whole evaluation
step of the reference's validation loop -- forward (lift) -> SplattingCUDA colour + depth + query x class logit maps -> lifting -- timed per
stage, in both precision modes.  (With seeded synthetic weights on uniform-noise images the outputs are chaotic functions of the input, so
scoring one mode against the other in PSNR / PQ says nothing about a trained network; those numbers need a checkpoint: evaluate.py.)
python tools/config3.py > gpurun_out/config3.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import synthetic
from siu3r_amd import synthetic_weights as OW
from siu3r_amd.gaussian_renderer import SplattingCUDA, lift_query_class_logits
from siu3r_amd.model import SIU3RModel

B, S, NV = int(os.environ.get("CFG3_B", "8")), 512, 6
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
images = torch.rand(B, 2, 3, S, S, generator=g).to(dev)
K = synthetic.default_intrinsics()[None, None].repeat(B, 2, 1, 1).to(dev)
ext = torch.stack([synthetic.target_views(NV, seed=10 * b) for b in range(B)])
Kt = synthetic.default_intrinsics()[None, None].repeat(B, NV, 1, 1)
sd = OW.make_weights(0)
rend = SplattingCUDA()


def sync_ms(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def run(precision):
    model = SIU3RModel(sd, image_size=(S, S), precision=precision, device=dev)
    fwd = lambda: model(images, K, enable_query_class_logit_lift=True)
    for _ in range(2):
        fwd()
    ms_fwd, out = sync_ms(fwd)
    gauss, seg, masks, infos, qs = out
    fresh = lambda: gauss.map_tensors(lambda t: t.clone())
    ms_col, r_col = sync_ms(lambda: rend.forward(fresh(), ext, Kt, (S, S), render_color=True))
    ms_qc, r_qc = sync_ms(lambda: rend.forward(fresh(), ext, Kt, (S, S), render_color=False, render_qc_logits=True))
    ms_lift, lifted = sync_ms(lambda: lift_query_class_logits(r_qc["render_qc_logits"], qs, num_queries=model.mask2former.num_queries,
                                                              label_ids_to_fuse=sorted(model.label_ids_to_fuse)))
    q = [t.shape[1] for t in r_qc["render_qc_logits"]]
    res = dict(precision=precision, forward_ms=ms_fwd, pairs_per_s_forward=B / ms_fwd * 1e3, render_color_ms_per_frame=ms_col / (B * NV),
               render_qc_logits_ms_per_frame=ms_qc / (B * NV), kept_queries=q, qc_channels=[21 * x for x in q], lifting_ms=ms_lift,
               step_ms=ms_fwd + ms_col + ms_qc + ms_lift, pairs_per_s_full_step=B / (ms_fwd + ms_col + ms_qc + ms_lift) * 1e3,
               segments=[len(i) for i in infos])
    del model
    torch.cuda.empty_cache()
    return res


rx = run("bf16x3")
rb = run("bf16")
print(json.dumps({"config": f"B={B} pairs 2x{S}x{S}, {NV} target views per pair, synthetic weights / images", "bf16x3": rx, "bf16": rb}))
