"""Checkpoint loader (reference inference.py:119-121, src/pipeline.py:25-39, src/models/model.py:116-176,
src/utils/weight_modify.py:163-228, src/models/backbone_croco.py:106-113): a Lightning-shaped .ckpt whose `hyper_parameters` pickle a
class that cannot be imported here, a MASt3R-shaped release .pth and the panoptic pre-training .ckpt.  CPU only."""
import dataclasses
import sys
import types

import pytest
import torch

from siu3r_amd import checkpoint as ck
from siu3r_amd import synthetic_weights as OW


def _small_sd():
    """a handful of real parameter names / shapes of the schema (the full 655 M-parameter dict is not needed to test the plumbing)"""
    spec = OW.param_spec()
    names = ["backbone.patch_embed.proj.weight", "backbone.enc_blocks.0.attn.qkv.weight", "backbone.dec_blocks.0.attn.qkv.weight",
             "backbone.dec_blocks2.0.attn.qkv.weight", "backbone.dec_blocks.3.mlp.fc1.bias", "backbone.dec_blocks2.3.mlp.fc1.bias",
             "downstream_head1.dpt.head.4.weight", "downstream_head1.dpt.head.4.bias", "adapter.spm.stem.1.num_batches_tracked",
             "mask2former.class_predictor.weight", "mask2former.model.transformer_module.queries_embedder.weight"]
    return {n: OW.make_tensor(n, spec[n], 0) for n in names}


def _write_lightning_ckpt(path, sd):
    """what Lightning writes for `Pipeline`: model.* + metric modules in state_dict, a pickled config object in hyper_parameters"""
    mod = types.ModuleType("fake_src_config")

    @dataclasses.dataclass
    class RootCfg:
        mode: str
        output_path: object
        nested: dict

    RootCfg.__module__ = "fake_src_config"
    RootCfg.__qualname__ = "RootCfg"
    mod.RootCfg = RootCfg
    sys.modules["fake_src_config"] = mod
    try:
        import pathlib

        state = {"model." + k: v for k, v in sd.items()}
        state["lpips.net.slice1.0.weight"] = torch.zeros(4, 3, 3, 3)
        state["psnr.total"] = torch.zeros(())
        torch.save({"epoch": 100, "global_step": 12345, "pytorch-lightning_version": "2.5.0", "state_dict": state,
                    "hyper_parameters": {"cfg": RootCfg("val", pathlib.PurePosixPath("outputs/x"), {"a": [RootCfg("t", None, {})]})},
                    "optimizer_states": [], "lr_schedulers": []}, path)
    finally:
        del sys.modules["fake_src_config"]


def test_lightning_ckpt_loads_without_the_pickled_config_class(tmp_path):
    sd = _small_sd()
    p = tmp_path / "siu3r_epoch100.ckpt"
    _write_lightning_ckpt(p, sd)
    assert "fake_src_config" not in sys.modules
    with pytest.raises(Exception):  # the stock loaders need the class (or refuse it)
        torch.load(p, map_location="cpu", weights_only=False)
    out = ck.load_siu3r_state_dict(p, verbose=False)
    assert "fake_src_config" not in sys.modules
    assert set(out) == set(sd)  # model. stripped, lpips.* / psnr.* dropped
    for k, v in sd.items():
        assert torch.equal(out[k], v)
    missing, unexpected = ck.check_keys(out)
    assert unexpected == [] and len(missing) > 1000 and "backbone.enc_norm.weight" in missing


def test_lpips_network_comes_out_of_a_pipeline_checkpoint(tmp_path):
    """Pipeline keeps its LPIPS metric as a sub-module (src/pipeline.py:35), so a Lightning checkpoint carries the VGG16 + lin tensors under
    `lpips.net.`: load_lpips_weights finds them through the restricted unpickler; a checkpoint without them gives None, a partial set an error"""
    from oracle import lpips_oracle as LO
    from siu3r_amd.lpips import VGG_SLICES

    lw = LO.random_weights(4)
    state = {"model." + k: v for k, v in _small_sd().items()}
    for k, sl in enumerate(VGG_SLICES):
        for i in sl:
            for q in ("weight", "bias"):
                state[f"lpips.net.net.slice{k + 1}.{i}.{q}"] = lw[f"features.{i}.{q}"]
    for k in range(5):
        state[f"lpips.net.lin{k}.model.1.weight"] = state[f"lpips.net.lins.{k}.model.1.weight"] = lw[f"lin{k}.model.1.weight"]
    p = tmp_path / "with_lpips.ckpt"
    torch.save({"state_dict": state, "pytorch-lightning_version": "2.5.0"}, p)
    got = ck.load_lpips_weights(p)
    assert torch.equal(got["conv28.weight"], lw["features.28.weight"]) and torch.equal(got["lin4"], lw["lin4.model.1.weight"].reshape(-1))
    assert not any(k.startswith("lpips") for k in ck.load_siu3r_state_dict(p, verbose=False))
    q = tmp_path / "model_only.ckpt"
    torch.save({"state_dict": {k: v for k, v in state.items() if k.startswith("model.")}}, q)
    assert ck.load_lpips_weights(q) is None
    r = tmp_path / "partial.ckpt"
    _write_lightning_ckpt(r, _small_sd())   # (carries one stray lpips tensor)
    with pytest.raises(RuntimeError, match="incomplete"):
        ck.load_lpips_weights(r)


def test_bare_state_dict_and_model_prefix(tmp_path):
    sd = _small_sd()
    p = tmp_path / "bare.pt"
    torch.save(sd, p)
    out = ck.load_siu3r_state_dict(p, verbose=False)
    assert set(out) == set(sd)
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, p)
    assert set(ck.load_siu3r_state_dict(p, verbose=False)) == set(sd)


def test_decoder_duplication_only_when_absent():
    sd = {k: v for k, v in _small_sd().items() if "dec_blocks2" not in k}
    out = ck.duplicate_decoder(sd)
    assert torch.equal(out["backbone.dec_blocks2.0.attn.qkv.weight"], sd["backbone.dec_blocks.0.attn.qkv.weight"])
    assert torch.equal(out["backbone.dec_blocks2.3.mlp.fc1.bias"], sd["backbone.dec_blocks.3.mlp.fc1.bias"])
    full = _small_sd()
    full["backbone.dec_blocks2.0.attn.qkv.weight"] = full["backbone.dec_blocks2.0.attn.qkv.weight"] + 1.0
    assert torch.equal(ck.duplicate_decoder(full)["backbone.dec_blocks2.0.attn.qkv.weight"], full["backbone.dec_blocks2.0.attn.qkv.weight"])


def test_mast3r_release_surgery(tmp_path):
    """weight_modify.py:163-228: `backbone.` prefix except the downstream heads, 4-channel pts3d+conf head cut to xyz, old flat patch
    embedding reshaped, dec_blocks duplicated."""
    g = torch.Generator().manual_seed(0)
    rel = {"patch_embed.proj.weight": torch.randn(1024, 768, generator=g), "patch_embed.proj.bias": torch.randn(1024, generator=g),
           "enc_blocks.0.attn.qkv.weight": torch.randn(8, 8, generator=g), "dec_blocks.0.attn.qkv.weight": torch.randn(8, 8, generator=g),
           "downstream_head1.dpt.head.4.weight": torch.randn(4, 128, 1, 1, generator=g), "downstream_head1.dpt.head.4.bias": torch.randn(4, generator=g),
           "downstream_head2.dpt.head.4.weight": torch.randn(4, 128, 1, 1, generator=g), "downstream_head2.dpt.head.4.bias": torch.randn(4, generator=g)}
    p = tmp_path / "MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric.pth"
    torch.save({"model": rel, "args": "Namespace(...)"}, p)
    out = ck.load_siu3r_state_dict(p, verbose=False)
    assert out["backbone.patch_embed.proj.weight"].shape == (1024, 3, 16, 16)
    assert torch.equal(out["backbone.patch_embed.proj.weight"].reshape(1024, -1), rel["patch_embed.proj.weight"])
    assert torch.equal(out["backbone.dec_blocks2.0.attn.qkv.weight"], rel["dec_blocks.0.attn.qkv.weight"])
    assert out["downstream_head1.dpt.head.4.weight"].shape == (3, 128, 1, 1) and torch.equal(out["downstream_head2.dpt.head.4.bias"], rel["downstream_head2.dpt.head.4.bias"][:3])
    assert "backbone.downstream_head1.dpt.head.4.weight" not in out and "backbone.enc_blocks.0.attn.qkv.weight" in out
    with pytest.raises(RuntimeError):
        ck.mast3r_to_siu3r({"patch_embed.proj.weight": torch.zeros(1024, 3, 14, 14)})


def test_seg_pretrain_surgery():
    """model.py:142-171: class predictor / criterion / backbone dropped, learned queries copied into a 100-row table."""
    sd = {"model.mask2former.class_predictor.weight": torch.ones(134, 256), "model.criterion.empty_weight": torch.ones(134),
          "model.backbone.blocks.0.w": torch.ones(2), "model.adapter.level_embed": torch.ones(3, 1024),
          "model.mask2former.model.transformer_module.queries_embedder.weight": torch.arange(50 * 256, dtype=torch.float32).view(50, 256),
          "model.mask2former.model.transformer_module.queries_features.weight": torch.ones(200, 256)}
    out = ck.seg_pretrain_to_siu3r(sd, num_queries=100)
    assert set(out) == {"adapter.level_embed", "mask2former.model.transformer_module.queries_embedder.weight", "mask2former.model.transformer_module.queries_features.weight"}
    qe = out["mask2former.model.transformer_module.queries_embedder.weight"]
    assert qe.shape == (100, 256) and torch.equal(qe[:50], sd["model.mask2former.model.transformer_module.queries_embedder.weight"]) and float(qe[50:].abs().sum()) == 0
    assert out["mask2former.model.transformer_module.queries_features.weight"].shape == (100, 256)


def test_expected_keys_match_the_generator():
    keys = ck.expected_keys()
    assert len(keys) > 1600 and "backbone.intrinsic_encoder.weight" in keys
    missing, unexpected = ck.check_keys({k: None for k in keys})
    assert missing == [] and unexpected == []


class _Evil:
    """pickles as REDUCE(builtins.eval, ("...",)): what a crafted checkpoint would use to run code at load time"""

    def __reduce__(self):
        import builtins

        return (builtins.eval, ("__import__('os').environ.__setitem__('SIU3R_PWNED', '1')",))


@pytest.mark.parametrize("payload", ["eval", "getattr", "hub"])
def test_crafted_checkpoint_cannot_execute_code(tmp_path, payload):
    """the fallback unpickler reconstructs an explicit allow-list only.  builtins.eval / getattr, torch.hub.load, numpy.load
    ... come back as inert placeholders; the weights beside them still load."""
    import os
    import pickle

    os.environ.pop("SIU3R_PWNED", None)

    class _G:
        def __init__(self, fn, args):
            self.fn, self.args = fn, args

        def __reduce__(self):
            return (self.fn, self.args)

    import builtins

    import numpy as np

    evil = {"eval": _Evil(), "getattr": _G(builtins.getattr, ("abc", "upper")), "hub": _G(torch.hub.load, ("x/y", "z"))}[payload]
    p = tmp_path / "evil.ckpt"
    # a class weights_only refuses, so that the loader takes its fallback path, next to the payload
    sd = _small_sd()
    _write_lightning_ckpt(p, sd)
    obj = torch.load(p, map_location="cpu", weights_only=False, pickle_module=ck._PickleModule)
    obj["hyper_parameters"]["evil"] = evil
    obj["callbacks"] = {"np": _G(np.load, ("/etc/passwd",))}
    torch.save(obj, p)
    out = ck.load_siu3r_state_dict(p, verbose=False)
    assert os.environ.get("SIU3R_PWNED") is None, "the payload ran"
    assert set(out) >= set(sd) and all(torch.equal(out[k], sd[k]) for k in sd)
    raw = ck.read_checkpoint_file(p)
    assert isinstance(raw["hyper_parameters"]["evil"], ck._Opaque) and isinstance(raw["callbacks"]["np"], ck._Opaque)
    # and the allow-list is explicit: nothing else under builtins / torch / numpy resolves
    for mod, name in (("builtins", "eval"), ("builtins", "exec"), ("builtins", "__import__"), ("builtins", "getattr"), ("torch.hub", "load"),
                      ("numpy", "load"), ("os", "system"), ("copyreg", "_reconstructor")):
        assert ck._Unpickler.find_class(ck._Unpickler.__new__(ck._Unpickler), mod, name) is ck._Opaque, (mod, name)
    # non-pickle failures are not swallowed into the fallback
    bad = tmp_path / "truncated.ckpt"
    bad.write_bytes(p.read_bytes()[:200])
    with pytest.raises(Exception):
        ck.read_checkpoint_file(bad)
