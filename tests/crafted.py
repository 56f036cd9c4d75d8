"""Crafted Mask2Former head outputs that exercise every branch of the panoptic post-process
(keep threshold, overlap-ratio test, stuff fusing, empty item, kept-but-none-accepted item).
Shared by the oracle pin tests, the golden generator and the GPU parity tests."""
import torch


def crafted_panoptic_inputs(Q=100, T=2, h=32, w=32, C=21, seed=0):
    g = torch.Generator().manual_seed(seed)
    B = 4
    cls = torch.full((B, Q, C), -4.0)
    cls[:, :, C - 1] = 4.0  # default: void with high confidence -> dropped
    msk = torch.randn(B, Q, T, h, w, generator=g) * 0.5 - 6.0  # default: confidently background
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")

    def blob(b, q, t, cy, cx, r, amp=8.0):
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        msk[b, q, t] = torch.where(d2 <= r * r, torch.tensor(amp), msk[b, q, t])

    def klass(b, q, c, logit=6.0):
        cls[b, q] = -4.0
        cls[b, q, c] = logit

    # item 0: two wall (stuff 0) queries, one floor (stuff 1), two chairs (thing 4) one of which loses the
    # overlap test against a stronger overlapping query, one low-score query, one void query
    klass(0, 3, 0); blob(0, 3, 0, 8, 8, 6); blob(0, 3, 1, 8, 8, 5)
    klass(0, 10, 0); blob(0, 10, 0, 24, 24, 5)
    klass(0, 11, 1); blob(0, 11, 1, 24, 8, 6)
    klass(0, 20, 4, logit=9.0); blob(0, 20, 0, 8, 24, 6, amp=10.0)
    klass(0, 21, 4, logit=5.0)
    msk[0, 21] = -12.0  # never wins the background ...
    blob(0, 21, 0, 9, 25, 6, amp=6.0)  # ... and is mostly covered by query 20 -> area/orig < 0.8 -> rejected
    klass(0, 30, 7, logit=-3.5)  # score below the keep threshold
    blob(0, 30, 0, 16, 16, 4)
    # item 1: nothing kept (empty branch, float -1 map)
    # item 2: kept queries whose masks never win / never reach 0.5 -> no accepted segment (quirk branch)
    klass(2, 5, 6, logit=3.0)
    # item 3: after an empty item the stale (height,width) are the target size (quirk), plus a thing-only scene
    klass(3, 7, 12); blob(3, 7, 0, 16, 16, 10); blob(3, 7, 1, 10, 10, 7)
    klass(3, 8, 13); blob(3, 8, 1, 24, 24, 5)
    return cls, msk


def permuted(cls, msk, order):
    """Re-order the batch items (the reference's stale (height, width) quirk depends on the order)."""
    idx = torch.tensor(order)
    return cls[idx].clone(), msk[idx].clone()
