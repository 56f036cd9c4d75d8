#!/usr/bin/env python
"""SIU3R hot-path benchmark (driver contract: one JSON line on rank 0).

A "step" is one pass of the hot path (SIU3RModel.forward incl. on-device panoptic post-process and
query-class-logit lifting inputs) over one batch of synthetic image pairs at 2 x 512 x 512 (BASELINE.json
configs[1]: single pair, ViT-L encoder/decoder + DPT 3DGS heads + ViT-Adapter/Mask2Former, bf16), with
seeded synthetic weights of the reference architecture (no checkpoint is available offline) and inputs
already resident in HBM.  value = image-pairs/s over all ranks (weak scaling: each rank runs its own pairs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_PAIR_512 = 4059.0e9  # algorithmic 2*MAC of one pair @512^2 (torch flop counter on the reference)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="image pairs per step per GPU (configs[1] = 1)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    args = ap.parse_args()

    from siu3r_amd import distributed as D
    from siu3r_amd import ops
    from siu3r_amd.model import SIU3RModel
    from siu3r_amd import synthetic_weights as OW  # shared synthetic-weight generator (no checkpoint offline)

    rank, local, world = D.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()  # (lets a gloo dry run put two ranks on one GPU; one GPU per rank otherwise)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    H = W = args.size
    B = args.batch

    sd = OW.make_weights(0)
    model = SIU3RModel(sd, image_size=(H, W), precision=args.precision, device=dev)
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.rand(B, 2, 3, H, W, generator=g).to(dev)
    K = torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, 2, 1, 1).to(dev)

    def step():
        with torch.no_grad():
            return model(images, K, enable_query_class_logit_lift=True)

    # untimed warm-up: at least 3 passes whatever W is (1st packs the weights eagerly, 2nd captures the HIP graphs, 3rd replays)
    for _ in range(max(3, args.warmup)):
        out = step()
    model.release_source_weights()
    del sd
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)

    # the path's single collective: per-rank additive statistics
    gauss = out[0]
    stats = dict(n_pairs=B * args.steps, n_images=2 * B * args.steps, n_gaussians=gauss.means.shape[1] * B,
                 n_segments=sum(len(i) for i in out[3]), label_checksum=float(gauss.instance_labels.sum().item()))
    gathered = D.all_gather_stats(D.pack_stats(stats), device=dev)
    total = D.reduce_stats(gathered)

    if rank != 0:
        return
    pairs = total["n_pairs"]
    value = pairs / dt
    result = {
        "metric": "image-pairs/sec @2x512^2 (SIU3R network forward incl. panoptic post-process)",
        "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.precision == "bf16" else "bf16x3 (fp32 activations, 3-pass bf16 MFMA)",
        "data": "synthetic (seeded uniform images, seeded synthetic weights of the reference architecture)",
        "config": {"workload": f"configs[1]: single pair 2x{H}x{W} per step" if B == 1 else f"{B} pairs 2x{H}x{W} per step",
                   "pairs_per_step_per_gpu": B, "image_size": [H, W], "precision": args.precision,
                   "parallelism": f"dp{world} (independent pairs, one all-gather of metric statistics)",
                   "launch": "per-chain HIP graphs on 6 streams" if (model.use_graph and model._ctx.concurrent) else "eager"},
        "network_tflops_algorithmic": value * FLOPS_PER_PAIR_512 * (H * W / (512 * 512)) / 1e12,
    }

    if not args.no_roofline:
        timer = ops.KernelTimer()
        ops.set_kernel_timer(timer)
        conc, model._ctx.concurrent = model._ctx.concurrent, False  # one stream: every launch between its own two events
        torch.cuda._sleep(int(2.0e8))  # ~0.1 s: the whole step is enqueued before it runs, so the event pairs time the
        step()                         # kernels back to back instead of the host's launch pace
        model._ctx.concurrent = conc
        ops.set_kernel_timer(None)
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_pass.sh: separate
        # FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md, both in KiB)
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc_path) and args.precision == "bf16" and B == 1 and (H, W) == (512, 512):
            pmc = json.load(open(pmc_path)).get("siu3r_gemm_dma::gemm_dma_kernel<1, 0, false, 2>") if name == "gemm_dma_kernel<1,0,false,2>" else None
            if pmc:
                traffic = (2.0 * pmc["fetch_size_per_launch"] + pmc["write_size_per_launch"]) * 1024.0
                traffic_src = "profiles/r01_pmc_summary.json (gemm_dma_kernel<1,0,false,2>, mean per launch)"
        result["roofline"] = {
            "kernel": f"siu3r_gemm_dma::{name} (dense Linear launches; HIP events around every launch of one eager single-stream step, queued behind a sleep kernel, after the timed region)",
            "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_step": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
            "algorithmic_flops_per_launch_avg": d["flops"] / d["launches"],
            "gemm_time_ms_per_step": d["ms"],
            "all_variants": {k: {"launches": v["launches"], "ms": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in summ.items()},
        }

    if not args.no_render:
        # render leg (the metric's "render ms/frame"): the pair's 524 288 Gaussians -> 6 target views @512^2 through the
        # K2-semantics path (SplattingCUDA.forward, colour + depth), timed with HIP events on the launch stream
        import copy
        from siu3r_amd import raster, synthetic
        from siu3r_amd.gaussian_renderer import SplattingCUDA
        from siu3r_amd.gaussians_types import Gaussians

        nv = 6
        ext = synthetic.target_views(nv)[None].repeat(B, 1, 1, 1)
        Kt = synthetic.default_intrinsics()[None, None].repeat(B, nv, 1, 1)
        rend = SplattingCUDA()
        def fresh():
            return Gaussians(means=gauss.means.clone(), covariances=gauss.covariances.clone(), harmonics=gauss.harmonics, opacities=gauss.opacities)
        rend.forward(fresh(), ext, Kt, (H, W), render_color=True)  # warm-up
        reps = 3
        gs = [fresh() for _ in range(reps)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for g_ in gs:
            rend.forward(g_, ext, Kt, (H, W), render_color=True)
        e1.record()
        torch.cuda.synchronize()
        ms_frame = e0.elapsed_time(e1) / (reps * B * nv)
        # data-dependent sizes of one view, for the algorithmic byte count
        from siu3r_amd import cuda_splatting as cs
        g1 = fresh()
        raster.scale_inplace_(g1.means, 10.0); raster.scale_inplace_(g1.covariances, 100.0)
        e = ext[0].clone(); e[:, :3, 3] *= 10.0
        _, _, aux = cs.render_cuda(e[1:2], Kt[0, 1:2], torch.tensor([1.0]), torch.tensor([1000.0]), (H, W), torch.zeros(1, 3), g1.means[:1], g1.covariances[:1], g1.harmonics[:1], g1.opacities[:1], return_aux=True)
        st = aux[0]["state"]
        G = g1.means.shape[1]; G_v = int((st["tiles_touched"] > 0).sum()); Dp = int(st["D"]); P = H * W
        bytes_alg = raster.algorithmic_bytes(G, G_v, Dp, P)
        result["render"] = {"ms_per_frame": ms_frame, "views": nv, "resolution": [H, W], "gaussians": G, "visible": G_v, "tile_pairs": Dp,
                            "semantics": "K2 (diff-gaussian-rasterization family): SH deg 4 -> RGB + depth + opacity + n_touched",
                            "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms_frame * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                         "frac": bytes_alg / (ms_frame * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_view": bytes_alg, "traffic": None,
                                         "note": "whole per-view pipeline (project+scan+fill+sort+composite, incl. host-side camera prep and the pair-count sync)"}}

    if not args.no_render and world == 1:
        # rasterizer stress leg (BASELINE.json configs[4] shape): 2 097 152 Gaussians (what 8 views x 512^2 produce), most of
        # them in view, one 1920x1080 frame, SH degree 4 -> RGB + depth (K2 semantics).  The network's own output with
        # synthetic weights leaves ~6 % of the Gaussians in view, which says little about the rasterizer's bandwidth
        import math
        from siu3r_amd import cuda_splatting as cs
        Gs, Ws, Hs = 2_097_152, 1920, 1080
        m_, cov_, op_, sh_ = (t.to(dev) for t in synthetic.random_scene(Gs, seed=1, spread=3.0, depth=(2.0, 9.0), scale=(0.004, 0.03)))
        cov6 = raster.cov6_from_cov3x3(cov_)
        shs = sh_.permute(0, 2, 1).contiguous()
        c2w = synthetic.perturbed_camera(0, jitter=0.1)
        w2c = torch.linalg.inv(c2w)
        fx = 0.9 * Ws
        fovx, fovy = 2 * math.atan(Ws / (2 * fx)), 2 * math.atan(Hs / (2 * fx))
        proj = cs.get_projection_matrix(torch.tensor([0.1]), torch.tensor([100.0]), torch.tensor([fovx]), torch.tensor([fovy]))[0]
        cam2 = raster.make_cam_k2(w2c=w2c, full_proj=proj @ w2c, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2), campos=c2w[:3, 3],
                                  bg=torch.zeros(3), width=Ws, height=Hs, sh_degree=4)
        for _ in range(2):
            o = raster.rasterize_k2(cam2, m_, cov6, shs, op_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = raster.rasterize_k2(cam2, m_, cov6, shs, op_)
        e1.record()
        torch.cuda.synchronize()
        ms_s = e0.elapsed_time(e1) / 5
        Gv_s, D_s = int((o["state"]["tiles_touched"] > 0).sum()), int(o["state"]["D"])
        b_s = raster.algorithmic_bytes(Gs, Gv_s, D_s, Hs * Ws)
        result["render_stress"] = {"ms_per_frame": ms_s, "resolution": [Ws, Hs], "gaussians": Gs, "visible": Gv_s, "tile_pairs": D_s,
                                   "scene": "siu3r_amd.synthetic.random_scene(seed=1): configs[4] shape (8 views x 512^2 worth of Gaussians, 1080p)",
                                   "roofline": {"bound": "hbm", "achieved": b_s / (ms_s * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                "frac": b_s / (ms_s * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_view": b_s, "traffic": None}}
        del m_, cov_, op_, sh_, cov6, shs, o

    if world == 1 and not args.no_cpu_baseline:
        from oracle import siu3r_oracle as O

        # bounded CPU sample: 16 threads (256 oversubscribed threads made one forward take 830 s on the GPU box);
        # a quarter-size pair first, and the full-size pair only if it is predicted to stay within ~40 s
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        sd_cpu = OW.make_weights(0)
        K_c = K[:1].cpu()
        img_q = torch.nn.functional.interpolate(images[0].cpu(), size=(H // 2, W // 2), mode="bilinear")[None]
        t1 = time.perf_counter()
        with torch.no_grad():
            O.model_forward(sd_cpu, img_q, K_c, keep_intermediates=False)
        t_q = time.perf_counter() - t1
        if t_q * 4.4 <= 40.0:
            t1 = time.perf_counter()
            with torch.no_grad():
                O.model_forward(sd_cpu, images[:1].cpu(), K_c, keep_intermediates=False)
            t_cpu = time.perf_counter() - t1
            sample = f"1 pair 2x{H}x{W}, one fp32 forward = {t_cpu:.1f} s"
        else:
            t_cpu = t_q * 4.27  # FLOP ratio 4059.0 / 950.7 GFLOP between 512^2 and 256^2
            sample = f"1 pair 2x{H//2}x{W//2} = {t_q:.1f} s, scaled x4.27 (FLOP ratio) to 2x{H}x{W}"
        result["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "image-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": sample + "; oracle/siu3r_oracle.py = parity-pinned fp32 port of the reference forward"}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
