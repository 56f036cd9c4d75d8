"""PLY export with the reference's vertex schema (reference src/utils/ply_export.py:28-97): x,y,z,nx,ny,nz(0),
f_dc_0..2, f_rest_* (channel-major, omitted when save_sh_dc_only), opacity (as is), scale_0..2 = log(scale),
rot_0..3 = wxyz of the RAW quaternion, semantic_label/instance_label (int32), seg_query_class_logits_0..q*c-1.
Binary little-endian, written with numpy only (the reference uses plyfile, which produces the same header grammar).
Host-side IO: not part of the GPU hot path."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def construct_list_of_attributes(num_rest: int) -> list:
    attrs = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(num_rest)]
    attrs += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return attrs + ["semantic_label", "instance_label"]


def export_ply(means, scales, rotations, harmonics, opacities, semantic_labels, instance_labels, seg_query_class_logits, path,
               shift_and_scale: bool = False, save_sh_dc_only: bool = True):
    path = Path(path)
    c = lambda t: t.detach().cpu()
    means, scales, rotations, harmonics, opacities = c(means).float(), c(scales).float(), c(rotations).float(), c(harmonics).float(), c(opacities).float()
    if shift_and_scale:
        means = means - means.median(dim=0).values
        sf = means.abs().quantile(0.95, dim=0).max()
        means, scales = means / sf, scales / sf
    x, y, z, w = rotations.numpy().T
    rot_wxyz = np.stack((w, x, y, z), axis=-1)
    f_dc = harmonics[..., 0].numpy()
    f_rest = harmonics[..., 1:].flatten(start_dim=1).numpy()
    names = construct_list_of_attributes(0 if save_sh_dc_only else f_rest.shape[1])
    dtype = [(n, "<f4") for n in names[:-2]]
    cols = [means.numpy(), np.zeros_like(means.numpy()), f_dc] + ([] if save_sh_dc_only else [f_rest]) + [opacities.numpy()[:, None], scales.log().numpy(), rot_wxyz]
    has_labels = semantic_labels is not None and instance_labels is not None
    if has_labels:
        dtype += [("semantic_label", "<i4"), ("instance_label", "<i4")]
    qcl = None
    if seg_query_class_logits is not None:
        g, q, cc = seg_query_class_logits.shape
        qcl = c(seg_query_class_logits).float().reshape(g, q * cc).numpy()
        dtype += [(f"seg_query_class_logits_{i}", "<f4") for i in range(q * cc)]
    n = means.shape[0]
    el = np.empty(n, dtype=dtype)
    flat = np.concatenate(cols, axis=1)
    for i, name in enumerate(names[:-2]):
        el[name] = flat[:, i]
    if has_labels:
        el["semantic_label"] = c(semantic_labels).numpy().astype(np.int32)
        el["instance_label"] = c(instance_labels).numpy().astype(np.int32)
    if qcl is not None:
        for i in range(qcl.shape[1]):
            el[f"seg_query_class_logits_{i}"] = qcl[:, i]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property {'int' if t == '<i4' else 'float'} {name}" for name, t in dtype]
    header += ["end_header"]
    path.parent.mkdir(exist_ok=True, parents=True)
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(el.tobytes())
    return path


def read_ply_vertices(path):
    """Minimal reader for the files written above (tests / viewer hand-off)."""
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"ply" and b"binary_little_endian" in fh.readline()
        n = int(fh.readline().split()[-1])
        dtype = []
        while True:
            line = fh.readline().decode().strip()
            if line == "end_header":
                break
            _, t, name = line.split()
            dtype.append((name, "<i4" if t == "int" else "<f4"))
        return np.frombuffer(fh.read(), dtype=dtype, count=n)
