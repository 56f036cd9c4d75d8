"""siu3r_amd: MI355X-native SIU3R inference path (hand-written HIP kernels behind a C ABI, see include/siu3r_hip.h).

Public names mirror the reference's (src/models/*): SIU3RModel, SIU3RMultiViewModel, AsymmetricCroCo, AsymmetricCroCoMulti,
CroCoViTAdapter, VideoMask2FormerForVideoSegmentation, UnifiedGaussianAdapter, SplattingCUDA, Gaussians, export_ply.
They are imported lazily: `import siu3r_amd` itself needs neither a GPU nor the built library."""
_LAZY = {
    "SIU3RModel": "model", "SIU3RMultiViewModel": "model", "AsymmetricCroCo": "model", "AsymmetricCroCoMulti": "model",
    "CroCoViTAdapter": "model", "VideoMask2FormerForVideoSegmentation": "model", "UnifiedGaussianAdapter": "model",
    "SplattingCUDA": "gaussian_renderer", "rasterize_splats": "gaussian_renderer", "Gaussians": "gaussians_types",
    "export_ply": "ply_export", "MetricAccumulator": "metrics",
}
__all__ = sorted(_LAZY)


def __getattr__(name):
    if name in _LAZY:
        import importlib

        return getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
