"""Integer-exact parity of the on-device panoptic post-process against the CPU oracle on crafted inputs
(keep threshold, overlap-ratio rejection, stuff fusing, empty item, kept-but-none-accepted item, and the
reference's stale (height, width) quirk)."""
import pytest
import torch

from crafted import crafted_panoptic_inputs, permuted

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", [[0, 1, 2, 3], [2, 0, 3, 1]], ids=["natural", "quirk-first"])
@pytest.mark.parametrize("target", [(64, 64), (96, 80)], ids=["64", "96x80"])
def test_panoptic_postprocess_exact(order, target):
    from oracle import siu3r_oracle as O
    from siu3r_amd.postprocess import VideoMask2FormerImageProcessor, post_process_gaussians
    from siu3r_amd.gaussians_types import Gaussians

    cls, msk = permuted(*crafted_panoptic_inputs(), order)
    ref = O.panoptic_postprocess(cls, msk, target)
    out = dict(class_queries_logits=cls.cuda(), masks_queries_logits=msk.cuda())
    res = VideoMask2FormerImageProcessor().post_process_panoptic_segmentation(
        out, threshold=0.5, target_sizes=[target] * len(order), label_ids_to_fuse={0, 1})
    assert len(res) == len(ref)
    for b, (r, m) in enumerate(zip(ref, res)):
        assert m["segmentation"].dtype == r["segmentation"].dtype, (b, m["segmentation"].dtype)
        assert torch.equal(m["segmentation"].cpu(), r["segmentation"]), f"segmentation differs for item {b}"
        # ids / labels / fused flags are integer outputs: exact.  Scores are fp32 softmax values rounded to 6 decimals
        # by the reference (:1444): compare within 2e-6 (the last printed digit may round differently).
        strip = lambda segs: [{k: v for k, v in s_.items() if k != "score"} for s_ in segs]
        assert strip(m["segments_info"]) == strip(r["segments_info"]), (b, m["segments_info"], r["segments_info"])
        assert all(abs(x["score"] - y["score"]) <= 2e-6 for x, y in zip(m["segments_info"], r["segments_info"]))
        assert len(m["query_scores"]) == len(r["query_scores"]) and all(abs(x - y) <= 2e-6 for x, y in zip(m["query_scores"], r["query_scores"]))
        assert tuple(m["query_class_logits"].shape) == tuple(r["query_class_logits"].shape), (b, m["query_class_logits"].shape, r["query_class_logits"].shape)
        err = float((m["query_class_logits"].cpu() - r["query_class_logits"]).abs().max())
        print(f"[parity] query_class_logits item {b}: max abs err {err:.2e}")
        assert err <= 2e-6
    B, (H, W) = len(order), target
    sem_ref, ins_ref = O.scatter_labels(ref, B, 2, H, W)
    g = Gaussians(means=torch.zeros(B, 2, H * W, 3), covariances=torch.zeros(B, 2, H * W, 3, 3), harmonics=torch.zeros(B, 2, H * W, 3, 25),
                  opacities=torch.zeros(B, 2, H * W), scales=torch.zeros(B, 2, H * W, 3), rotations=torch.zeros(B, 2, H * W, 4))
    g, masks, infos, qcl, qs = post_process_gaussians(g, res, B, 2, H, W, True)
    assert torch.equal(g.semantic_labels.cpu(), sem_ref) and torch.equal(g.instance_labels.cpu(), ins_ref)
    assert g.semantic_labels.dtype == torch.int32 and g.means.shape == (B, 2 * H * W, 3)


def _network_like_logits(Q=100, T=2, h=128, w=128, C=21, kept=14, seed=0):
    """class / mask logits with the statistics of the network's own output at 2 x 512^2 (mask size 128^2): `kept` confident non-void
    queries (two stuff classes among them), smooth mask logits around -2.8 with unit-scale variation, so that segments border each other."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(1, Q, C, generator=g)
    cls[:, :, C - 1] += 9.0
    lab = [1, 1, 0, 5, 12, 19, 13, 1, 10, 3, 19, 7, 4, 15][:kept]
    for i, c in enumerate(lab):
        cls[0, 7 * i + 2, c] += 14.0
    coarse = torch.randn(1 * Q * T, 1, h // 8, w // 8, generator=g)
    msk = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).view(1, Q, T, h, w) * 2.2 - 2.8
    return cls, msk + 0.05 * torch.randn(1, Q, T, h, w, generator=g)


@pytest.mark.parametrize("target", [(512, 512), (256, 256)], ids=["512", "256"])
def test_panoptic_postprocess_full_size(target):
    """The kept-query kernels (argmax over kept queries at the target size, area / original-area counts, stuff fusing, the
    [T*H*W, q, 21] query x class logit volume) at the benchmark's size on network-like logits: integer outputs bit-exact against the
    oracle ON IDENTICAL INPUT LOGITS."""
    from oracle import siu3r_oracle as O
    from siu3r_amd.postprocess import VideoMask2FormerImageProcessor

    cls, msk = _network_like_logits()
    ref = O.panoptic_postprocess(cls, msk, target)[0]
    assert len(ref["segments_info"]) >= 5 and any(s_["was_fused"] for s_ in ref["segments_info"])
    out = dict(class_queries_logits=cls.cuda(), masks_queries_logits=msk.cuda())
    res = VideoMask2FormerImageProcessor().post_process_panoptic_segmentation(out, threshold=0.5, target_sizes=[target], label_ids_to_fuse={0, 1})[0]
    strip = lambda segs: [(s_["id"], s_["label_id"], s_["was_fused"]) for s_ in segs]
    assert strip(res["segments_info"]) == strip(ref["segments_info"])
    assert all(abs(x["score"] - y["score"]) <= 2e-6 for x, y in zip(res["segments_info"], ref["segments_info"]))
    same = float((res["segmentation"].cpu() == ref["segmentation"]).float().mean())
    print(f"[parity] full-size segmentation {target}: {len(ref['segments_info'])} segments, identical pixels {same:.7f}")
    assert torch.equal(res["segmentation"].cpu(), ref["segmentation"]), f"segmentation differs on {1 - same:.2e} of the pixels"
    err = float((res["query_class_logits"].cpu() - ref["query_class_logits"]).abs().max())
    assert tuple(res["query_class_logits"].shape) == tuple(ref["query_class_logits"].shape) and err <= 2e-6, err



def _old_layout_volume(pend, B, T, Q, MS=256):
    """the stage's probability planes [B, T, Q, MS, MS] (kept queries only, query-major: round 6) as the channel-last volume
    [B, T, MS, MS, Q] of rounds 2-5, which the round-3/4 code objects index"""
    import torch
    planes, kept, tab = pend["p256"], pend["kept_idx"], pend["tab"]
    nk = tab[5 * B * Q:5 * B * Q + B]
    vol = torch.zeros((B, T, MS, MS, Q), dtype=torch.float32, device=planes.device)
    for b in range(B):
        n = int(nk[b])
        vol[b][..., kept[b, :n].long()] = planes[b, :, :n].permute(0, 2, 3, 1)
    return vol


def test_panoptic_stage_beside_bf16_mfma_waves_is_run_to_run_identical():
    """Round 6 regression of the round-4 panoptic label flake: the device stage on one stream (B = 8, soft segment borders: wide bands
    where two queries compete) while a second stream keeps two bf16-MFMA-issuing waves on every SIMD (tools/probes/pk_hazard/burn.hip,
    burn_wave: one-wave workgroups that leave registers and wave slots for the stage's workgroups); 100 launches must give identical
    label / segment maps.  Background: a packed fp32 instruction with op_sel:[x,1] -- what the SLP
    vectoriser made of the argmax's lerp in rounds 3-5 -- returns wrong low halves beside bf16 MFMAs on gfx950 (tools/probes/pk_hazard/xwave2.hip);
    the shipped kernels hold no packed arithmetic.  INFORMATIONAL CONTROL: the round-3/4 code object (`e0`, tools/probes/pk_hazard/gen.py)
    is launched eight times per iteration on the stage's buffers and its wrong maps are counted and printed -- it failed in 25 % of the
    network's forwards and in 3-6 % of its launches beside the SLP-built bf16x3 128 x 64 GEMM, but which neighbours share a SIMD with it
    often enough is a property of their footprints (0 beside this synthetic neighbour and beside the rebuilt GEMM), so the count is not
    asserted."""
    import ctypes, os, subprocess, sys
    from siu3r_amd import ops
    from siu3r_amd.postprocess import VideoMask2FormerImageProcessor

    B, T, Q, C, h, w, H, W = 8, 2, 100, 21, 128, 128, 512, 512
    g = torch.Generator().manual_seed(31)
    low = torch.randn(B * T, Q, 12, 12, generator=g)
    msk = (torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False) * 2.0).view(B, T, Q, h, w).permute(0, 2, 1, 3, 4).contiguous()
    cls = torch.randn(B, Q, C, generator=g)
    cls[:, :25, :-1] += 4.0 * torch.nn.functional.one_hot(torch.randint(0, C - 1, (B, 25), generator=g), C - 1)  # 25 confident queries per item
    cls[:, 25:, -1] += 6.0                                                                                   # the rest: void
    out = dict(class_queries_logits=cls.cuda(), masks_queries_logits=msk.cuda())
    proc = VideoMask2FormerImageProcessor()
    # the control code object (needs hipcc + the LLVM tools of the ROCm image; without them the control is skipped, not the test)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e0 = burn = None
    sink = torch.zeros(4096, device="cuda")
    try:
        build = os.path.join(root, "tools", "probes", "pk_hazard", "_build")
        if not os.path.exists(os.path.join(build, "ppa_e0.hsaco")):
            subprocess.check_call([sys.executable, os.path.join(root, "tools", "probes", "pk_hazard", "gen.py")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 6 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(build, "ppa_e0.hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"ppa_e0") == 0
        e0 = fn
        bmod, bfn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(bmod), os.path.join(build, "burn.hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(bfn), bmod, b"burn_wave") == 0
        burn = bfn
    except Exception as ex:  # pragma: no cover
        print(f"[hazard control] unavailable: {ex}")
    NC = 8  # control launches per iteration (behind the stage, back to back, as tools/probes/pk_hazard/standalone.py runs them)
    lab0 = torch.zeros(NC, B, T, H, W, dtype=torch.int32, device="cuda")
    scr = torch.zeros(8192, dtype=torch.int32, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ref, bad, ctl_wrong, prev, vol_old = None, [], 0, None, None
    try:
        for it in range(100):
            if it:  # (the first launch runs alone: the reference maps)
                if burn is not None:  # two MFMA-issuing waves on every SIMD for ~10 ms
                    bv = [ctypes.c_void_p(sink.data_ptr()), ctypes.c_int(60000)]
                    ba = (ctypes.c_void_p * 2)(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in bv])
                    assert hip.hipModuleLaunchKernel(burn, 2048, 1, 1, 64, 1, 1, 0, ctypes.c_void_p(sb.cuda_stream), ba, None) == 0
            def control(src, c0, c1):  # launches c0 .. c1 - 1 of the control on the buffers of the stage result `src` (volume: the old layout)
                for c in range(c0, c1):
                    vals = [ctypes.c_void_p(vol_old.data_ptr()), ctypes.c_void_p(src["keep"][2].data_ptr()), ctypes.c_void_p(src["kept_idx"].data_ptr()),
                            ctypes.c_void_p(src["tab"].data_ptr() + 4 * 5 * B * Q), ctypes.c_void_p(lab0[c].data_ptr()), ctypes.c_void_p(scr.data_ptr()), ctypes.c_void_p(scr.data_ptr() + 16384),
                            ctypes.c_int(T), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(256), ctypes.c_int(Q), ctypes.c_float(0.5)]
                    arr = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
                    assert hip.hipModuleLaunchKernel(e0, (T * H * W + 255) // 256, B, 1, 256, 1, 1, 0, ctypes.c_void_p(sa.cuda_stream), arr, None) == 0
            if e0 is not None and vol_old is None:  # (the inputs are the same every iteration: converted once, from an untimed stage run)
                vol_old = _old_layout_volume(proc.begin_panoptic(out, threshold=0.5, target_sizes=[(H, W)] * B, label_ids_to_fuse={0, 1}), B, T, Q)
                torch.cuda.synchronize()
            with torch.cuda.stream(sa):
                if e0 is not None and prev is not None:
                    control(prev, 0, NC // 2)  # ahead of the stage, on the previous (identical, complete) buffers: beside the first GEMM launches
                pend = proc.begin_panoptic(out, threshold=0.5, target_sizes=[(H, W)] * B, label_ids_to_fuse={0, 1})
                if e0 is not None:
                    control(pend, NC // 2 if prev is not None else 0, NC)
            prev = pend
            torch.cuda.synchronize()
            lab = pend["keep"][4].clone()
            seg = pend["seg"].clone()
            if ref is None:
                ref = (lab, seg)
                assert int(pend["tab"][5 * B * Q:5 * B * Q + B].min()) >= 10, "the crafted logits keep at least 10 queries per item"
                assert e0 is None or all(torch.equal(lab0[c], lab) for c in range(NC)), "the control computes the same map on an idle chip"
            else:
                if not (torch.equal(lab, ref[0]) and torch.equal(seg, ref[1])):
                    bad.append((it, int((lab != ref[0]).sum()), int((seg != ref[1]).sum())))
                ctl_wrong += 0 if e0 is None else sum(not torch.equal(lab0[c], ref[0]) for c in range(NC))
    finally:
        pass
    print(f"[hazard control] the round-3/4 code object gave a wrong label map in {ctl_wrong} of {99 * NC} launches beside MFMA-issuing waves; the shipped kernel in {len(bad)}")
    assert not bad, bad
