import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from siu3r_amd.model import SIU3RModel
from siu3r_amd import synthetic_weights as OW
dev = torch.device("cuda", 0)
sd = OW.make_weights(0)
m = SIU3RModel(sd, image_size=(512, 512), precision="bf16", device=dev)
g = torch.Generator().manual_seed(0)
imgs = [torch.rand(1, 2, 3, 512, 512, generator=g).to(dev) for _ in range(2)]
K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(1, 2, 1, 1).to(dev)
m.use_graph = False
ref = [m(i, K, enable_query_class_logit_lift=True) for i in imgs]
ref = [(r[0].means.clone(), r[0].harmonics.clone(), r[1].class_queries_logits.clone(), r[0].instance_labels.clone(), r[3]) for r in ref]
m.use_graph = True
for rep in range(3):
    for i, im in enumerate(imgs):
        o = m(im, K, enable_query_class_logit_lift=True)
        d = [(o[0].means - ref[i][0]).abs().max().item(), (o[0].harmonics - ref[i][1]).abs().max().item(),
             (o[1].class_queries_logits - ref[i][2]).abs().max().item(), (o[0].instance_labels != ref[i][3]).sum().item(), o[3] == ref[i][4]]
        print(rep, i, d)
for mode in (True, False):
    m.use_graph = mode
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(imgs[0], K, enable_query_class_logit_lift=True)
    torch.cuda.synchronize(); print("graph" if mode else "eager", (time.perf_counter() - t0) * 100, "ms/step")
m._ctx.concurrent = False
m.use_graph = True
m._graphs.clear()
for _ in range(3): m(imgs[0], K, enable_query_class_logit_lift=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m(imgs[0], K, enable_query_class_logit_lift=True)
torch.cuda.synchronize(); print("graph single-stream", (time.perf_counter() - t0) * 100, "ms/step")
m.use_graph = False
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m(imgs[0], K, enable_query_class_logit_lift=True)
torch.cuda.synchronize(); print("eager single-stream", (time.perf_counter() - t0) * 100, "ms/step")
