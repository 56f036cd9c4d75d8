"""Panoptic post-processing + Gaussian label scatter, on device.

Mirror of ``VideoMask2FormerImageProcessor.post_process_panoptic_segmentation`` (reference
src/models/mask2former/image_processing_video_mask2former.py:1238-1481) and of
``SIU3RModel.post_process_gaussians`` (reference src/models/model.py:231-312).  The integer maps are
produced by the kernels in csrc/postprocess.hip; the host only reads the small per-query segment table
back (one sync) to build the ``segments_info`` lists the reference returns as Python objects.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Set, Tuple

import torch

from . import _lib
from ._lib import check
from .ops import _gpu, _p, _stream

MASK_SIZE = 256  # hard-coded in the reference (image_processing_video_mask2former.py:1298)


class VideoMask2FormerImageProcessor:
    def post_process_panoptic_segmentation(self, outputs, threshold: float = 0.5, mask_threshold: float = 0.5,
                                           overlap_mask_area_threshold: float = 0.8,
                                           label_ids_to_fuse: Optional[Set[int]] = None,
                                           target_sizes: Optional[List[Tuple[int, int]]] = None,
                                           word_embeddings=None, with_query_class_logits: bool = True):
        pend = self.begin_panoptic(outputs, threshold, mask_threshold, overlap_mask_area_threshold, label_ids_to_fuse, target_sizes, word_embeddings)
        return self.finish_panoptic(pend, with_query_class_logits)

    def begin_panoptic(self, outputs, threshold: float = 0.5, mask_threshold: float = 0.5, overlap_mask_area_threshold: float = 0.8,
                       label_ids_to_fuse: Optional[Set[int]] = None, target_sizes: Optional[List[Tuple[int, int]]] = None, word_embeddings=None,
                       host_slot: int = 0):
        """First half: the device stage (probabilities, 256^2 masks, argmax maps, segment table) and an asynchronous copy of the small
        tables into pinned host memory, all on the CURRENT stream; returns the pending state.  SIU3RModel enqueues this right behind
        Mask2Former on the segmentation stream, so that it runs beside the heads instead of after them.  The host copy of the tables
        is valid until the next begin_panoptic() with the same host_slot."""
        assert word_embeddings is None, "the refer head has no executable reference behaviour (SURVEY.md A.17)"
        if label_ids_to_fuse is None:
            label_ids_to_fuse = set()
        class_logits = outputs["class_queries_logits"]
        mcl = outputs.get("_masks_channel_last") if isinstance(outputs, dict) else None
        if mcl is None:  # [B,Q,T,h,w] -> channel-last copy
            mcl = outputs["masks_queries_logits"].permute(0, 2, 3, 4, 1).contiguous()
        _gpu(class_logits, mcl)
        class_logits = class_logits.contiguous().float()
        B, Q, Cc = class_logits.shape
        _, T, IH, IW, _ = mcl.shape
        assert target_sizes is not None and all(tuple(t) == tuple(target_sizes[0]) for t in target_sizes)
        H, W = target_sizes[0]
        dev = class_logits.device
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        probs, scores, labels, kept_idx, n_keep = f32(B, Q, Cc), f32(B, Q), i32(B, Q), i32(B, Q), i32(B)
        p256 = f32(B, T, Q, MASK_SIZE, MASK_SIZE)  # planes of the kept queries (k < n_keep[b]); the rest stays unwritten
        lab_map, area, orig = i32(B, T, H, W), i32(B, Q), i32(B, Q)
        # the tables the host reads, in ONE buffer: [seg_id | seg_label | seg_fused | acc_list | seg_score bits] x [B, Q], then n_keep, n_acc
        tab = i32(5 * B * Q + 2 * B)
        seg_id, seg_label, seg_fused, acc_list = (tab[i * B * Q:(i + 1) * B * Q].view(B, Q) for i in range(4))
        seg_score = tab[4 * B * Q:5 * B * Q].view(torch.float32).view(B, Q)
        n_keep_t, n_acc = tab[5 * B * Q:5 * B * Q + B], tab[5 * B * Q + B:]
        seg, sem, ins = i32(B, T, H, W), i32(B, T, H, W), i32(B, T, H, W)
        fuse_mask = 0
        for c in label_ids_to_fuse:
            assert 0 <= c < 32
            fuse_mask |= 1 << c
        check(_lib.lib().siu3r_panoptic_stage1(
            _p(class_logits), _p(mcl), _p(probs), _p(scores), _p(labels), _p(kept_idx), _p(n_keep_t), _p(p256), _p(lab_map),
            _p(area), _p(orig), _p(seg_id), _p(seg_label), _p(seg_fused), _p(seg_score), _p(acc_list), _p(n_acc), _p(seg),
            _p(sem), _p(ins), B, T, Q, Cc, IH, IW, H, W, MASK_SIZE, threshold, mask_threshold, overlap_mask_area_threshold,
            fuse_mask, _stream()))
        # one device->host read of the small tables (the reference does a .item() per query instead), asynchronous into pinned memory
        # (pinned staging buffer per host_slot: results that are pending at the same time -- SIU3RModel.forward_async -- must not share one)
        tabs = self.__dict__.setdefault("_host_tabs", {})
        host = tabs.get(host_slot)
        if host is None or host.numel() != tab.numel():
            host = tabs[host_slot] = torch.empty(tab.numel(), dtype=torch.int32, pin_memory=True)
        host.copy_(tab, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return dict(ev=ev, host=host, tab=tab, dims=(B, T, Q, Cc, IH, IW, H, W), probs=probs, kept_idx=kept_idx, acc_list=acc_list, p256=p256, seg=seg,
                    sem=sem, ins=ins, keep=(class_logits, mcl, scores, labels, lab_map, area, orig))

    def finish_panoptic(self, pend, with_query_class_logits: bool = True):
        """Second half: wait for the tables (an event, not a device sync), build the reference's `segments_info` lists and enqueue the
        query x class volume of every item on the current stream."""
        B, T, Q, Cc, IH, IW, H, W = pend["dims"]
        pend["ev"].synchronize()
        h = pend["host"]
        table = h[:4 * B * Q].view(4, B, Q)
        sc_h = h[4 * B * Q:5 * B * Q].view(torch.float32).view(B, Q)
        nk_h, na_h = h[5 * B * Q:5 * B * Q + B].tolist(), h[5 * B * Q + B:].tolist()
        probs, kept_idx, acc_list, p256, seg, sem, ins = (pend[k] for k in ("probs", "kept_idx", "acc_list", "p256", "seg", "sem", "ins"))
        dev = seg.device
        results = []
        height, width = IH, IW  # the reference's stale (height, width) of the mask logits (quirk, :1297/:1355/:1470)
        for b in range(B):
            if nk_h[b] == 0:
                height, width = H, W
                segmentation = torch.full((T, H, W), -1.0, dtype=torch.float32, device=dev)
                qcl = None
                if with_query_class_logits:
                    qcl = torch.zeros((T * H * W, 1, Cc), dtype=torch.float32, device=dev)
                    qcl[:, 0, -1] = 1
                results.append(dict(segmentation=segmentation, segments_info=[], query_scores=[0.0], _qcl_gaussian_major=qcl,
                                    _qcl_hw=(H, W), _semantic=sem[b], _instance=ins[b]))
                continue
            segs, q_scores = [], []
            for j in range(na_h[b]):
                k = int(table[3, b, j])
                s_ = round(float(sc_h[b, k]), 6)
                segs.append(dict(id=int(table[0, b, k]), label_id=int(table[1, b, k]), was_fused=bool(table[2, b, k]), score=s_))
                q_scores.append(s_)
            qcl, qhw = None, (H, W)
            if with_query_class_logits:
                if na_h[b] > 0:
                    qcl = torch.empty((T * H * W, na_h[b], Cc), dtype=torch.float32, device=dev)
                    check(_lib.lib().siu3r_panoptic_qcl(_p(p256), _p(probs), _p(kept_idx), _p(acc_list), na_h[b], _p(qcl), b, T,
                                                        H, W, MASK_SIZE, Q, Cc, _stream()))
                else:
                    qhw = (height, width)
                    qcl = torch.zeros((T * height * width, 1, Cc), dtype=torch.float32, device=dev)
                    qcl[:, 0, -1] = 1
            results.append(dict(segmentation=seg[b], segments_info=segs, query_scores=q_scores, _qcl_gaussian_major=qcl,
                                _qcl_hw=qhw, _semantic=sem[b], _instance=ins[b]))
        for r in results:  # reference layout [T, q, C+1, H, W] as a view of the Gaussian-major buffer
            g = r["_qcl_gaussian_major"]
            if g is not None:
                hh, ww = r["_qcl_hw"]
                r["query_class_logits"] = g.view(T, hh, ww, g.shape[1], Cc).permute(0, 3, 4, 1, 2)
        return results


def post_process_gaussians(gaussians, results, B, V, H, W, enable_query_class_logit_lift=False):
    """SIU3RModel.post_process_gaussians (model.py:247-312): labels onto Gaussians, flatten views."""
    masks = [r["segmentation"] for r in results]
    infos = [r["segments_info"] for r in results]
    gaussians.semantic_labels = torch.stack([r["_semantic"] for r in results]).reshape(B, V * H * W)
    gaussians.instance_labels = torch.stack([r["_instance"] for r in results]).reshape(B, V * H * W)
    qcl = qscores = None
    if enable_query_class_logit_lift:
        qcl = [r["_qcl_gaussian_major"] for r in results]  # 'n q c h w -> (n h w) q c' (model.py:261-263)
        qscores = [r["query_scores"] for r in results]
        gaussians.seg_query_class_logits = qcl
    gaussians.means = gaussians.means.reshape(B, V * H * W, 3)
    gaussians.covariances = gaussians.covariances.reshape(B, V * H * W, 3, 3)
    gaussians.scales = gaussians.scales.reshape(B, V * H * W, 3)
    gaussians.rotations = gaussians.rotations.reshape(B, V * H * W, 4)
    gaussians.opacities = gaussians.opacities.reshape(B, V * H * W)
    gaussians.harmonics = gaussians.harmonics.reshape(B, V * H * W, 3, -1)
    return gaussians, masks, infos, qcl, qscores
