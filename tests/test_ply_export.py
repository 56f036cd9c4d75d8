"""CPU test of the PLY writer's schema (reference src/utils/ply_export.py:12-25, 61-97)."""
import numpy as np
import torch

from siu3r_amd.ply_export import construct_list_of_attributes, export_ply, read_ply_vertices


def test_ply_schema_roundtrip(tmp_path):
    G, q, c = 50, 2, 21
    g = torch.Generator().manual_seed(0)
    means, scales = torch.randn(G, 3, generator=g), torch.rand(G, 3, generator=g) * 0.01 + 1e-4
    rot, sh, op = torch.randn(G, 4, generator=g), torch.randn(G, 3, 25, generator=g), torch.rand(G, generator=g)
    sem, ins = torch.randint(0, 21, (G,), dtype=torch.int32), torch.randint(0, 5, (G,), dtype=torch.int32)
    qcl = torch.rand(G, q, c, generator=g)
    p = export_ply(means, scales, rot, sh, op, sem, ins, qcl, tmp_path / "o" / "output.ply", save_sh_dc_only=False)
    v = read_ply_vertices(p)
    names = list(v.dtype.names)
    assert names[: 6 + 3 + 72 + 1 + 3 + 4 + 2] == construct_list_of_attributes(72)
    assert names[-1] == f"seg_query_class_logits_{q * c - 1}" and len(v) == G
    assert np.allclose(v["x"], means[:, 0]) and np.all(v["nx"] == 0)
    assert np.allclose(v["f_dc_1"], sh[:, 1, 0]) and np.allclose(v["f_rest_24"], sh[:, 1, 1])  # channel-major rest
    assert np.allclose(v["opacity"], op) and np.allclose(v["scale_2"], scales[:, 2].log())
    assert np.allclose(v["rot_0"], rot[:, 3]) and np.allclose(v["rot_1"], rot[:, 0])             # wxyz of the raw xyzw
    assert v["semantic_label"].dtype == np.int32 and np.array_equal(v["instance_label"], ins.numpy())
    assert np.allclose(v["seg_query_class_logits_22"], qcl[:, 1, 1])
    v2 = read_ply_vertices(export_ply(means, scales, rot, sh, op, sem, ins, None, tmp_path / "dc.ply", save_sh_dc_only=True))
    assert list(v2.dtype.names) == construct_list_of_attributes(0)
