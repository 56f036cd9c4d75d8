"""Checkpoint loading for the MI355X path (reference: inference.py:119-121 `Pipeline.load_from_checkpoint(path, map_location="cpu",
strict=False)`; src/pipeline.py:25-39 (the LightningModule pickles its `RootCfg` into `hyper_parameters`); src/models/model.py:116-176
(`load_recon_ckpt` / `load_dust3r_ckpt` / `load_seg_ckpt`); src/utils/weight_modify.py:163-228 (`checkpoint_filter_fn`);
src/models/backbone_croco.py:106-113 (dec_blocks -> dec_blocks2 duplication)).

What a file may be:
  * a Lightning `.ckpt` of `Pipeline` / `PipelineMultiView`: {"state_dict": {"model.backbone...": ..., "lpips.net...": ...},
    "hyper_parameters": {"cfg": <pickled src.config.RootCfg>}, ...}.  Unpickling `hyper_parameters` needs the reference's `src`
    package, Hydra and dacite; none of it is needed to run the network, so unknown classes are replaced by an inert placeholder
    while unpickling (an explicit allow-list of tensor-rebuild functions, storages and plain containers is reconstructed for real);
  * a MASt3R / DUSt3R release `.pth`: {"model": state_dict of AsymmetricMASt3R, "args": ...} -> `mast3r_to_siu3r`;
  * the panoptic pre-training `.ckpt` ({"state_dict": {"model.adapter...", "model.mask2former..."}}) -> `seg_pretrain_to_siu3r`;
  * a bare state dict.
Host-side code: no GPU work, nothing here touches oracle/."""
from __future__ import annotations

import pickle
from typing import Dict, Iterable, List, Optional, Tuple

import torch


class _Opaque:
    """Stand-in for any class the checkpoint pickled that is not a tensor container (configs, paths, enums, callbacks)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):  # e.g. enum lookups pickled as Class(value)
        return _Opaque()


# Globals the fallback unpickler reconstructs for real: exactly what a tensor container needs.  Everything else -- including
# builtins.eval / exec / getattr / __import__, torch.hub.*, numpy.load -- becomes an inert placeholder, so a crafted file cannot REDUCE
# its way to code execution.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "bytearray"), ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._utils", "_rebuild_qtensor"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"),
    ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"),
}
_ALLOWED_TORCH_SUFFIXES = ("Storage",)  # torch.FloatStorage, torch.storage.UntypedStorage, ... (typed storages of legacy files)


def _allowed(module: str, name: str) -> bool:
    if (module, name) in _ALLOWED_GLOBALS:
        return True
    if module in ("torch", "torch.storage") and name.endswith(_ALLOWED_TORCH_SUFFIXES):
        return True
    if module == "torch" and name in {str(d).split(".")[-1] for d in (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64,
                                                                        torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool)}:
        return True
    return False


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if _allowed(module, name):
            return super().find_class(module, name)
        return _Opaque


class _PickleModule:
    """`pickle_module` argument of torch.load: the stock pickle with the class-filtering Unpickler."""
    __name__ = "siu3r_amd.checkpoint._PickleModule"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps, Pickler, HIGHEST_PROTOCOL = pickle.dump, pickle.dumps, pickle.Pickler, pickle.HIGHEST_PROTOCOL


def read_checkpoint_file(path) -> dict:
    """torch.load without importing whatever the file pickled besides tensors."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        # the file pickles classes besides tensors (a Lightning checkpoint's RootCfg): allow-listed unpickler, placeholders for the rest.
        # Any other failure (truncated zip, I/O) propagates.
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)


def extract_state_dict(ckpt) -> Tuple[Dict[str, torch.Tensor], str]:
    """-> (flat state dict, kind) with kind in {"lightning", "release", "bare"}."""
    if isinstance(ckpt, dict) and isinstance(ckpt.get("state_dict"), dict):
        return dict(ckpt["state_dict"]), "lightning"
    if isinstance(ckpt, dict) and isinstance(ckpt.get("model"), dict):
        return dict(ckpt["model"]), "release"
    if isinstance(ckpt, dict) and ckpt and all(isinstance(v, torch.Tensor) for v in ckpt.values()):
        return dict(ckpt), "bare"
    raise RuntimeError("unrecognised checkpoint: expected a Lightning .ckpt ('state_dict'), a MASt3R/DUSt3R .pth ('model') or a bare state dict")


def strip_pipeline_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Pipeline.state_dict() -> SIU3RModel.state_dict(): keep `model.*` (prefix removed), drop the metric modules
    (`lpips.*`, `psnr.*`, ...: src/pipeline.py:31-36).  A dict without any `model.` key is returned unchanged."""
    if not any(k.startswith("model.") for k in sd):
        return dict(sd)
    return {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}


def duplicate_decoder(sd: Dict[str, torch.Tensor], prefix: str = "backbone.") -> Dict[str, torch.Tensor]:
    """backbone_croco.py:106-113: a checkpoint without `dec_blocks2` gets a copy of `dec_blocks`."""
    out = dict(sd)
    if not any(k.startswith(prefix + "dec_blocks2") for k in sd):
        for k, v in sd.items():
            if k.startswith(prefix + "dec_blocks."):
                out[prefix + "dec_blocks2." + k[len(prefix + "dec_blocks."):]] = v
    return out


def mast3r_to_siu3r(sd: Dict[str, torch.Tensor], patch: Tuple[int, int] = (16, 16), in_chans: int = 3) -> Dict[str, torch.Tensor]:
    """`checkpoint_filter_fn` (weight_modify.py:163-228) for a MASt3R / DUSt3R state dict: patch-embed weights of the pre-conv format are
    reshaped to [O, I, 16, 16]; every key that is not a downstream head gets the `backbone.` prefix; the confidence channel of the
    pts3d heads is dropped (`head.4` weight / bias rows 0..2); `dec_blocks2` is duplicated from `dec_blocks` when absent.  Resampling
    a patch embedding of a different patch size / channel count (timm `resample_patch_embed`, `adapt_input_conv`) is not needed by
    any released weight file and is rejected here."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if "patch_embed.proj.weight" in k:
            if v.dim() < 4:
                v = v.reshape(v.shape[0], -1, patch[0], patch[1])
            if tuple(v.shape[1:]) != (in_chans, patch[0], patch[1]):
                raise RuntimeError(f"{k}: patch embedding {tuple(v.shape)} does not match ({in_chans}, {patch[0]}, {patch[1]})")
        out[k if "downstream_head" in k else "backbone." + k] = v
    for h in ("downstream_head1", "downstream_head2"):
        for p in ("weight", "bias"):
            key = f"{h}.dpt.head.4.{p}"
            if key in out:
                out[key] = out[key][0:3]
    return duplicate_decoder(out)


def seg_pretrain_to_siu3r(sd: Dict[str, torch.Tensor], num_queries: int = 100) -> Dict[str, torch.Tensor]:
    """`load_seg_ckpt` (model.py:142-171): drop the class predictor, the criterion and the pre-training backbone, strip `model.`,
    and copy the learned queries into the first rows of a `num_queries`-row table (rows beyond the file's are zero here; the
    reference leaves them at nn.Embedding's random initialisation)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if "class_predictor" in k or "criterion" in k or "backbone" in k:
            continue
        nk = k[len("model."):] if k.startswith("model.") else k
        if "queries_embedder" in k or "queries_features" in k:
            t = torch.zeros((num_queries, v.shape[1]), dtype=v.dtype)
            t[: min(num_queries, v.shape[0])] = v[:num_queries]
            v = t
        out[nk] = v
    return out


def expected_keys() -> List[str]:
    """state-dict keys of SIU3RModel (the synthetic-weight generator enumerates the same schema)."""
    from .synthetic_weights import param_spec

    return list(param_spec().keys())


def check_keys(sd: Dict[str, torch.Tensor], expected: Optional[Iterable[str]] = None) -> Tuple[List[str], List[str]]:
    """(missing, unexpected) like nn.Module.load_state_dict(strict=False).  BatchNorm `num_batches_tracked` counters are ignored."""
    exp = {k for k in (expected if expected is not None else expected_keys()) if not k.endswith("num_batches_tracked")}
    have = {k for k in sd if not k.endswith("num_batches_tracked")}
    return sorted(exp - have), sorted(have - exp)


def load_siu3r_state_dict(path, strict: bool = False, verbose: bool = True, ckpt=None) -> Dict[str, torch.Tensor]:
    """Everything `Pipeline.load_from_checkpoint(path, strict=False).model.state_dict()` would hold, as CPU tensors keyed by the
    reference's parameter names -- what `SIU3RModel(state_dict, ...)` takes.  ckpt: the file's content when the caller has already read it
    (`read_checkpoint_file(path)`: a Lightning checkpoint with optimizer state is multi-GB; evaluate.py also takes its LPIPS tensors from it)."""
    sd, kind = extract_state_dict(read_checkpoint_file(path) if ckpt is None else ckpt)
    if kind == "release":
        sd = mast3r_to_siu3r(sd)
    else:
        sd = strip_pipeline_prefix(sd)
        if any(k.startswith("backbone.dec_blocks.") for k in sd):
            sd = duplicate_decoder(sd)
    sd = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}
    missing, unexpected = check_keys(sd)
    if verbose and (missing or unexpected):
        print(f"[siu3r_amd.checkpoint] {path}: {len(missing)} missing / {len(unexpected)} unexpected keys"
              + (f" (first missing: {missing[:3]})" if missing else "") + (f" (first unexpected: {unexpected[:3]})" if unexpected else ""))
    if strict and (missing or unexpected):
        raise RuntimeError(f"{path}: missing keys {missing[:8]}..., unexpected keys {unexpected[:8]}...")
    return sd


def load_lpips_weights(path, ckpt=None) -> Optional[Dict[str, torch.Tensor]]:
    """The LPIPS (VGG16 + lin layers) tensors of a checkpoint file, or None when it holds none.  A Pipeline checkpoint carries them under
    `lpips.` (src/pipeline.py:35: the metric is a sub-module of the LightningModule); a state dict of the metric alone or of the `lpips`
    package's network is accepted too (siu3r_amd.lpips.weights_from_state_dict matches the key tails).  ckpt: the already-read content of
    `path` (see load_siu3r_state_dict)."""
    from .lpips import weights_from_state_dict

    if ckpt is None:
        ckpt = read_checkpoint_file(path)
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    if not isinstance(sd, dict):
        return None
    return weights_from_state_dict({k: v for k, v in sd.items() if isinstance(v, torch.Tensor)})
