"""Helpers shared by the golden-fixture tests (oracle on CPU, HIP path on the GPU box)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRIDE = 4093


def load_model_fixture(size):
    z = np.load(os.path.join(GOLDEN, f"model_{size}.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, f"model_{size}.json")))
    return z, meta


def fixture_images(size):
    """The fixture input: the reference's assets/living_room pair (decoded pixels committed as data in
    input_pair_u8.npz), /255, bilinearly resized exactly as tests/golden/make_golden.py does."""
    a = torch.from_numpy(np.load(os.path.join(GOLDEN, "input_pair_u8.npz"))["pair"]).permute(0, 3, 1, 2).float() / 255.0
    if a.shape[-1] != size:
        a = torch.stack([torch.nn.functional.interpolate(x[None], size=(size, size), mode="bilinear", align_corners=False)[0] for x in a])
    return a[None]


def default_K(B=1, V=2):
    return torch.tensor([[318 / 256, 0, 0.5], [0, 318 / 256, 0.5], [0, 0, 1]])[None, None].repeat(B, V, 1, 1)


def fixture_images_multi(size=128):
    """V = 3 fixture input: the asset pair and the first image mirrored (tests/golden/make_golden.py::multi_fixture)."""
    pair = fixture_images(size)[0]
    return torch.stack((pair[0], pair[1], pair[0].flip(-1)))[None]


def load_multi_fixture(V=3, size=128):
    z = np.load(os.path.join(GOLDEN, f"model_multi_v{V}_{size}.npz"), allow_pickle=False)
    return z, json.load(open(os.path.join(GOLDEN, f"model_multi_v{V}_{size}.json")))


def compare_summary(name, t: torch.Tensor, z, tol):
    f = t.detach().float().cpu().reshape(-1)
    want = torch.from_numpy(z[f"{name}.sample"])
    got = f[::STRIDE]
    assert list(t.shape) == list(z[f"{name}.shape"]), (name, t.shape, z[f"{name}.shape"])
    scale = float(z[f"{name}.absmax"]) + 1e-30
    err = float((got - want).abs().max()) / scale
    l2 = abs(float(f.double().norm()) - float(z[f"{name}.l2"])) / (float(z[f"{name}.l2"]) + 1e-30)
    print(f"[golden] {name:24s} sample max-normalised err {err:.3e}  |l2 rel diff| {l2:.3e}")
    assert err <= tol and l2 <= tol, (name, err, l2)
    return err


FIELDS = ("means", "covariances", "harmonics", "opacities", "scales", "rotations")
