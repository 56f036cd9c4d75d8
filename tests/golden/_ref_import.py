"""Import harness for the *reference* SIU3R python package (build container only).

This file is test infrastructure.  It registers five stub modules for packages
the image lacks (jaxtyping, lightning_utilities, hydra, omegaconf, dacite) so that
``/root/reference/src/models`` imports unmodified, exactly as SURVEY.md section 8(c)
describes.  Nothing here is imported by the product path, by ``-m gpu`` tests, by
``bench.py`` or by ``smoke()`` -- /root/reference does not exist on the GPU box.
Only ``tests/golden/make_golden.py`` and the container-only pin tests use it.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


def install_stubs() -> None:
    if "jaxtyping" not in sys.modules:
        jt = types.ModuleType("jaxtyping")

        class _Ann:
            def __class_getitem__(cls, item):
                return cls

        for name in ("Float", "Bool", "Int64", "Int", "Shaped", "UInt8", "Int32"):
            setattr(jt, name, type(name, (_Ann,), {}))
        sys.modules["jaxtyping"] = jt
    if "lightning_utilities" not in sys.modules:
        lu = types.ModuleType("lightning_utilities")
        core = types.ModuleType("lightning_utilities.core")
        rz = types.ModuleType("lightning_utilities.core.rank_zero")

        def rank_zero_only(fn):
            return fn

        rank_zero_only.rank = 0

        def rank_prefixed_message(msg, rank):
            return f"[rank: {rank}] {msg}"

        rz.rank_zero_only = rank_zero_only
        rz.rank_prefixed_message = rank_prefixed_message
        lu.core = core
        core.rank_zero = rz
        sys.modules["lightning_utilities"] = lu
        sys.modules["lightning_utilities.core"] = core
        sys.modules["lightning_utilities.core.rank_zero"] = rz
    if "hydra" not in sys.modules:
        sys.modules["hydra"] = types.ModuleType("hydra")
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.DictConfig = dict
        oc.OmegaConf = type("OmegaConf", (), {})
        oc.open_dict = lambda cfg: cfg
        sys.modules["omegaconf"] = oc
    if "dacite" not in sys.modules:
        dc = types.ModuleType("dacite")
        dc.from_dict = lambda *a, **k: None
        sys.modules["dacite"] = dc
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def build_reference_model(image_size=(256, 256), multi=False):
    """Construct the reference SIU3RModel (random init) on CPU in eval mode."""
    install_stubs()
    import torch
    from src.config import ModelCfg, CrocoCfg, Mask2formerCfg, GaussianHeadCfg
    from src.utils.scannet_constant import PANOPTIC_SEMANTIC2NAME, STUFF_CLASSES

    cfg = ModelCfg(
        croco=CrocoCfg(),
        mask2former=Mask2formerCfg(
            id2label=PANOPTIC_SEMANTIC2NAME, label_ids_to_fuse=STUFF_CLASSES
        ),
        gaussian_head=GaussianHeadCfg(),
        image_size=list(image_size),
    )
    if multi:
        from src.models.model_multi import SIU3RMultiViewModel as M
    else:
        from src.models.model import SIU3RModel as M
    with torch.no_grad():
        model = M(cfg)
    model.eval()
    return model
