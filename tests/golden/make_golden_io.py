"""Golden vectors for the host-side data path (build container only; needs /root/reference):
  scannet_gt.npz   seeded panoptic id maps of a 3-view group -> (mask_labels, class_labels) produced by the REFERENCE's
                   VideoMask2FormerImageProcessor.preprocess exactly as ScanNetDataset calls it (scannet_dataset.py:66-73, 283-288),
                   plus relative_pose / intrinsics_normalize outputs of the reference's dataset methods (:76-115).
Run:  python tests/golden/make_golden_io.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def panoptic_maps(seed=0, n=3, H=256, W=256):
    """semantic ids in 0..20, instance ids unique per (semantic, blob); id 0 = unlabelled; one view lacks a class the others have"""
    rng = np.random.default_rng(seed)
    sems, inss = [], []
    for v in range(n):
        sem = np.zeros((H, W), np.int64)
        ins = np.zeros((H, W), np.int64)
        # stuff: wall (1) on top, floor (2) at the bottom
        sem[: H // 4] = 1
        ins[: H // 4] = 1
        sem[3 * H // 4:] = 2
        ins[3 * H // 4:] = 2
        for k in range(4):  # things
            if v == 1 and k == 3:
                continue
            cls = [5, 7, 5, 12][k]
            y0, x0 = rng.integers(H // 4, H // 2), rng.integers(0, W - 70)
            sem[y0:y0 + 40, x0:x0 + 60] = cls
            ins[y0:y0 + 40, x0:x0 + 60] = 3 + k
        sems.append(sem)
        inss.append(ins)
    return sems, inss


def main():
    assert R.reference_available()
    R.install_stubs()
    from src.data.components.scannet_dataset import ScanNetDataset
    from src.models.mask2former import VideoMask2FormerImageProcessor

    sems, inss = panoptic_maps()
    proc = VideoMask2FormerImageProcessor(size=(256, 256), reduce_labels=True, do_rescale=False, do_normalize=False, ignore_index=255, num_labels=20)
    i2s = []
    for sem, ins in zip(sems, inss):  # scannet_dataset.py:263-281
        d = {}
        for s in np.unique(sem):
            for i in np.unique(ins[sem == s]):
                d[i] = s
        i2s.append(d)
    frames = [np.zeros((3,) + sems[0].shape, np.uint8) for _ in sems]
    enc = proc.preprocess(video_frames=frames, segmentation_maps=inss, instance_id_to_semantic_id=i2s, return_tensors="pt")
    ds = ScanNetDataset.__new__(ScanNetDataset)
    rng = np.random.default_rng(1)

    def pose():
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m = np.eye(4)
        m[:3, :3] = q * np.sign(np.linalg.det(q))
        m[:3, 3] = rng.normal(size=3)
        return m

    cext, text = [pose(), pose()], [pose(), pose(), pose()]
    rc, rt = ds.relative_pose(cext, text)
    K = np.array([[577.87, 0, 319.5, 0], [0, 577.87, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    Kn = ds.intrinsics_normalize([K])[0]
    np.savez_compressed(os.path.join(HERE, "scannet_gt.npz"), sem=np.stack(sems).astype(np.uint8), ins=np.stack(inss).astype(np.uint8), mask_labels=enc["mask_labels"].numpy().astype(np.uint8),
                        class_labels=enc["class_labels"].numpy(), cext=np.stack(cext), text=np.stack(text), rel_c=np.stack(rc), rel_t=np.stack(rt), K=K, Kn=Kn)
    print("mask_labels", tuple(enc["mask_labels"].shape), "class_labels", enc["class_labels"].tolist())


if __name__ == "__main__":
    main()
